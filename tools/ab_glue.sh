#!/bin/bash
# A/B of the torch-glue removals (round 5): gradient sink + parameter blocks; MetNet and ConvLSTM lines, alternating, same box.
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for cfg in "SF_NO_GRAD_SINK=1 SF_NO_PARAM_BLOCKS=1" "SF_NO_PARAM_BLOCKS=1" ""; do
    echo "== ${cfg:-default}"
    env $cfg python bench.py --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('metnet   %.1f samples/s %.3f ms' % (r['value'], r['ms_per_step']))"
    env $cfg python bench.py --workload convlstm --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('convlstm %.1f samples/s %.3f ms' % (r['value'], r['ms_per_step']))"
  done
done
