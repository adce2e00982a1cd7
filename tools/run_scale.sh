#!/bin/bash
# Scaling curve of bench.py on ONE node: N in {1,2,4,8} ranks (one per GPU, RCCL over xGMI), weak (8 samples/GPU, the default)
# and strong (BASELINE cfg 4: global batch 64 split over the ranks).  One JSON line per run under $OUT (default gpurun_out/scale).
#   tools/run_scale.sh [steps] [warmup]        e.g. tools/run_scale.sh 20 5
# Rendezvous on 127.0.0.1 (the container hostname may not resolve); HSA_ENABLE_IPC_MODE_LEGACY=0 is required for RCCL's dmabuf IPC.
set -u
cd "$(dirname "$0")/.."
STEPS=${1:-20}; WARM=${2:-5}
OUT=${OUT:-gpurun_out/scale}; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python -c 'import torch; print(torch.cuda.device_count())')
PORT=29500
for MODE in weak strong; do
  for N in 1 2 4 8; do
    [ "$N" -gt "$NGPU" ] && continue
    if [ "$MODE" = strong ]; then  # per-GPU batch 64 / N: skip a leg that cannot fit (about 2.7 GiB of HBM per MetNet sample in the benchmarked mode)
      FIT=$(python - "$N" <<'PY'
import sys, torch
free, _ = torch.cuda.mem_get_info(0)
print(int(free / 2**30 > 2.7 * 64 / int(sys.argv[1]) + 8))
PY
)
      if [ "$FIT" != 1 ]; then echo "strong N=$N: per-GPU batch $((64 / N)) does not fit this GPU's free memory - leg skipped"; continue; fi
    fi
    PORT=$((PORT + 1))
    ARGS="--gpus $N --steps $STEPS --warmup $WARM --scaling $MODE --no-cpu-baseline --no-extra"
    if [ "$N" -eq 1 ]; then
      python bench.py $ARGS > "$OUT/${MODE}_n$N.json" 2> "$OUT/${MODE}_n$N.err"
    else
      python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" bench.py $ARGS \
        > "$OUT/${MODE}_n$N.json" 2> "$OUT/${MODE}_n$N.err"
    fi
    python - "$OUT/${MODE}_n$N.json" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(f"{d['scaling']:6s} N={d['n_gpus']} batch/GPU={d['config']['per_gpu_batch']:3d}  {d['value']:9.1f} samples/s  {d['ms_per_step']:8.2f} ms/step  comm={d.get('comm')}")
except Exception as e:  # noqa: BLE001
    print("FAILED", sys.argv[1], e)
PY
  done
done
# ONE SCALE-shaped record of the whole sweep ($OUT/SCALE.json): per leg N, samples/s, ms/step, exposed communication, efficiency against the N = 1 leg of
# the same mode (weak: value_N / (N * value_1); strong: the same ratio - a fixed global batch N times faster is 1.0) - what the driver's SCALE_rNN.json
# holds, so that the first 8-GPU run needs no hand work.
python - "$OUT" <<'PY'
import glob, json, os, sys
out = sys.argv[1]
rec = {"what": "tools/run_scale.sh: bench.py at N = 1, 2, 4, 8 ranks of one node (RCCL over xGMI)", "legs": {}}
for mode in ("weak", "strong"):
    legs = []
    for f in sorted(glob.glob(os.path.join(out, f"{mode}_n*.json")), key=lambda p: int(p.rsplit("_n", 1)[1][:-5])):
        try:
            d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        except Exception as e:  # noqa: BLE001
            legs.append({"file": os.path.basename(f), "error": str(e)})
            continue
        comm = d.get("comm") or {}
        legs.append({"n_gpus": d["n_gpus"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "per_gpu_batch": d["config"]["per_gpu_batch"],
                     "global_batch": d["config"]["global_batch"], "exposed_comm_ms": comm.get("exposed_comm_ms"), "comm": comm or None, "dtype": d["dtype"]})
    base = next((l for l in legs if l.get("n_gpus") == 1), None)
    for l in legs:
        if base and "value" in l:
            l["efficiency_vs_n1"] = l["value"] / (l["n_gpus"] * base["value"])
    rec["legs"][mode] = legs
json.dump(rec, open(os.path.join(out, "SCALE.json"), "w"), indent=1)
print(json.dumps(rec))
PY
