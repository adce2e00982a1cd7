#!/bin/bash
# A/B of the gradient sink on the CloudGAN line (the one step that is half host-bound: fewer launches count there), alternating on one box.
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "SF_NO_GRAD_SINK=1" ""; do
  echo "== ${cfg:-default}"
  env $cfg python bench.py --workload cloudgan --steps 40 --warmup 20 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cloudgan %.1f samples/s %.3f ms' % (r['value'], r['ms_per_step']), r['config'].get('launch'))"
done; done
