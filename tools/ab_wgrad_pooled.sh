#!/bin/bash
# A/B of conv4's weight gradient forms inside the MetNet step (round 5), alternating on one box:
#   SF_NO_WGRAD_SPARSE=1 dense kernel | SF_NO_WGRAD_POOLED=1 2:4-sparse operand compressed from dout | default: operand from the pooled gradient + codes
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for cfg in "SF_NO_WGRAD_SPARSE=1" "SF_NO_WGRAD_POOLED=1" ""; do
    echo "== ${cfg:-default}"
    env $cfg python bench.py --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('metnet   %.1f samples/s %.3f ms' % (r['value'], r['ms_per_step']))"
  done
done
