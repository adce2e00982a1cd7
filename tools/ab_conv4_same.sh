#!/bin/bash
# A/B (round 5): the PatchGAN's 4x4 stride-2 convolutions on the unpadded space_to_depth2 form (default) against the padded input + cropped output
# (SF_CONV4_PADDED=1), CloudGAN line, alternating on one box.
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for cfg in "SF_CONV4_PADDED=1" ""; do
    echo "== ${cfg:-default}"
    env $cfg python bench.py --workload cloudgan --steps 40 --warmup 20 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cloudgan %.1f samples/s %.3f ms' % (r['value'], r['ms_per_step']))"
  done
done
