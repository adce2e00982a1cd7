"""Which weight-gradient launches does a MetNet bf16a step make, and how long does each take?  (GPU box: PYTHONPATH=. python tools/probe_wgrad_shapes.py)
Wraps satflow_amd.kernels.conv3x3_bwd_weight / _folded for ONE step and times each call with HIP events."""
import torch
import satflow_amd
from satflow_amd import kernels as K
import bench

satflow_amd.set_compute_dtype("bf16a")
dev = torch.device("cuda:0")
wl = bench.build_workload("metnet", dev, 8, 0)
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
log = []


def wrap(name):
    real = getattr(K, name)

    def f(*a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = real(*a, **kw); e1.record()
        ts = [x for x in a if hasattr(x, "c") and hasattr(x, "stride")]
        ints = [x for x in a if isinstance(x, int)]
        log.append((name, [(t.c, t.stride, t.dtype) for t in ts], ints[:3], e0, e1))
        return r
    setattr(K, name, f)


for nm in ("conv3x3_bwd_weight", "conv3x3_bwd_weight_folded"):
    wrap(nm)
wl.step()
torch.cuda.synchronize()
for name, ts, ints, e0, e1 in log:
    print(f"{e0.elapsed_time(e1) * 1e3:9.1f} us  {name:28s} tensors (c, stride, dtype) {ts}  n,h,w {ints}")
