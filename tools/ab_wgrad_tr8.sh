#!/bin/bash
# A/B (round 5): conv4's pooled sparse weight gradient on 8-row K tiles (default) against 4-row tiles (SF_WGRAD_TR8=0), MetNet step, alternating on one box.
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  for cfg in "SF_WGRAD_TR8=0" ""; do
    echo "== ${cfg:-default}"
    env $cfg python bench.py --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('metnet   %.1f samples/s %.3f ms' % (r['value'], r['ms_per_step']))"
  done
done
