#!/bin/bash
# usage (GPU box): tools/ab_lib.sh <baseline .so, in-tree> [pmc]  - same-box A/B of two builds of the library (SATFLOW_HIP_LIB picks the baseline): the MetNet
# bf16a step and kernel table, alternating, twice each; with `pmc`, then the HBM traffic of the step's kernels with the working-tree build.
set -u
BASE=$GRAFT_REPO_ROOT/$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/ab_lib; mkdir -p $OUT
ARGS="--steps 12 --warmup 4 --no-cpu-baseline --no-exchange-probe"
for i in 1 2; do
  SATFLOW_HIP_LIB=$BASE python3 bench.py $ARGS > $OUT/base_$i.json 2> $OUT/base_$i.err
  python3 bench.py $ARGS > $OUT/new_$i.json 2> $OUT/new_$i.err
done
python3 - <<'PY'
import json, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/ab_lib/"
for n in ("base_1", "new_1", "base_2", "new_2"):
    try:
        d = json.loads(open(out + n + ".json").read().strip().splitlines()[-1])
    except Exception as e:
        print(n, "failed", e, open(out + n + ".err").read()[-600:]); continue
    print(n, "ms_per_step", round(d["ms_per_step"], 3), "loss", d["extra"].get("final_loss"))
    for k in d["extra"]["kernels"]["rows"][:8]:
        print("     %-60s %s" % (k["kernel"][:60], {a: k[a] for a in k if a not in ("kernel", "replaces")}))
PY
if [ "${2:-}" = pmc ]; then
  bash tools/prof_pmc_step.sh ab_lib_pmc --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-exchange-probe
  python3 tools/pmc_by_kernel.py gpurun_out/ab_lib_pmc 8
fi
