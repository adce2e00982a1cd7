#!/bin/bash
# usage (GPU box): tools/ab_lib_quick.sh <baseline .so, in-tree> [bench args...]  - same-box A/B of two builds of the library on a bench line (default: MetNet
# bf16a without the extras), alternating, three runs each: ms_per_step (SATFLOW_HIP_LIB picks the baseline).
set -u
BASE=$GRAFT_REPO_ROOT/$1; shift
ARGS="${*:---steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-exchange-probe}"
one() { python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],3), "ms", round(d["value"],1))'; }
for i in 1 2 3; do
  echo "base: $(SATFLOW_HIP_LIB=$BASE python3 bench.py $ARGS 2>/dev/null | one)"
  echo "new:  $(python3 bench.py $ARGS 2>/dev/null | one)"
done
