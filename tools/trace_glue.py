"""Which Python lines launch the small torch kernels (fills, copies, adds, cats) of one training step?

    python tools/trace_glue.py metnet|convlstm|cloudgan|dgmr  [mode]

Runs three warm-up steps, profiles ONE step with torch.profiler (with_stack) and prints, per aten op that launched a device kernel, the call count,
the device time and the innermost frames inside this repository.  The output is the work list for "remove torch glue".
"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

import satflow_amd
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "metnet"
mode = sys.argv[2] if len(sys.argv) > 2 else "bf16a"
satflow_amd.set_compute_dtype(mode)
dev = torch.device("cuda:0")
if name == "metnet":
    wl = bench.MetNetWorkload(dev, 8, 0)
elif name == "convlstm":
    wl = bench.ConvLSTMWorkload(dev, 8, 0)
elif name == "cloudgan":
    wl = bench.CloudGANWorkload(dev, 8, 0)
else:
    os.environ["SF_NO_GRAPH"] = "1"
    wl = bench.DGMRWorkload(dev, 2, 0)
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    wl.step()
    torch.cuda.synchronize()

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if not ev.name.startswith("aten::") or not ev.kernels:
        continue
    frames = [f for f in (ev.stack or []) if root in f or "satflow_amd" in f or "bench.py" in f]
    where = " <- ".join(f.replace(root + "/", "").split(",")[0].strip() + ":" + f.split("line")[-1].strip().split()[0] if "line" in f else f for f in frames[:3])
    shapes = str([tuple(s) for s in (ev.input_shapes or []) if s])[:60]
    k = (ev.name, (where.split(' <- ')[0] + ' ' + shapes) if (where.startswith('bench.py') or os.environ.get('SF_GLUE_SHAPES')) else (where or shapes))
    agg[k][0] += 1
    agg[k][1] += sum(kk.duration for kk in ev.kernels)
tot_n = sum(v[0] for v in agg.values())
tot_t = sum(v[1] for v in agg.values())
print(f"{name} {mode}: {tot_n} aten ops with device kernels in one step, {tot_t:.0f} us of device time")
for (op, where), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:150]:
    print(f"{n:5d} {t:9.1f} us  {op:28s} {where}")
