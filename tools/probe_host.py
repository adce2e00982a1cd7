"""How long does the host need to ENQUEUE one training step (no sync)?  If this exceeds the GPU step time the run is host-bound."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import satflow_amd
from bench import MetNetWorkload, ConvLSTMWorkload
satflow_amd.set_compute_dtype("bf16")
dev = torch.device("cuda:0")
for name, cls in (("metnet", MetNetWorkload), ("convlstm", ConvLSTMWorkload)):
    wl = cls(dev, 8, 0)
    for _ in range(3): wl.step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); wl.step(); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t0))
    print(name, "enqueue ms", [round(a * 1e3, 1) for a, _ in ts], "total ms", [round(b * 1e3, 1) for _, b in ts])
    del wl
