#!/bin/bash
# A/B of the pooling in conv4's epilogue (SF_NO_POOL_FUSE=1: conv4 + maxpool kernel), one gpurun call: MetNet tests, then the step both ways, twice.
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_metnet_gpu.py tests/test_bf16a_gpu.py tests/test_litmetnet_gpu.py -q -m gpu -x 2>&1 | tail -3
for rep in 1 2; do
for v in "SF_NO_POOL_FUSE=1" "SF_POOL_FUSE=1"; do
  echo "== $v"
  env $v python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'samples/s', round(d['ms_per_step'],3), 'ms')"
done; done
bash tools/prof_stats.sh r05_metnet_bf16a_poolfuse --dtype bf16a --no-cpu-baseline --no-extra > /dev/null 2>&1
python - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05_metnet_bf16a_poolfuse/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:14]:
    print(f"{r['Name'][:95]:95s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1e3:9.1f}us")
PY
