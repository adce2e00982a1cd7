#!/bin/bash
# usage (GPU box): tools/ab_env.sh VAR=VALUE [bench args...]  - same-box A/B of an environment switch on a bench line (default: MetNet bf16a), alternating, three
# runs each: ms_per_step with and without the variable set.
set -u
KV=$1; shift
ARGS="${*:---steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-exchange-probe}"
for i in 1 2 3; do
  echo "default: $(python3 bench.py $ARGS 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],3), "ms", round(d["value"],1))')"
  echo "$KV: $(env $KV python3 bench.py $ARGS 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],3), "ms", round(d["value"],1))')"
done
