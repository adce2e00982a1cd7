"""Timing probe of the fused ConvLSTM cell (cfg 2: 8 images 128x128, 64 input lanes + 64 hidden) and its backward kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, satflow_amd
from satflow_amd._hip import T, NULL, gate_storage_dtype
from satflow_amd.models.layers.ConvLSTM import ConvLSTMCell
satflow_amd.set_compute_dtype(os.environ.get("SF_MODE", "bf16a"))
dev = torch.device("cuda:0")
n, H, W, cin, hid = 8, 128, 128, 64, 64
cell = ConvLSTMCell(cin, hid, (3, 3), True).to(dev)
eng = cell.engine
x = torch.randn(n, H, W, eng.cinp, device=dev)
h0 = torch.randn(n, H, W, eng.hidp, device=dev); c0 = torch.randn_like(h0)
h1 = torch.empty_like(h0); c1 = torch.empty_like(h0)
gates = torch.empty(n, H, W, 4 * eng.hidp, device=dev, dtype=gate_storage_dtype())
def timeit(f, iters=30):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
tag = os.environ.get("SATFLOW_HIP_LIB", "default")[-20:]
t = timeit(lambda: eng.step(T(x), h0, c0, n, H, W, h1, c1, gates))
fl = 2 * 9 * (eng.cinp + eng.hidp) * 4 * hid * H * W * n
print(f"{tag:>20} lstm cell fwd (gates {str(gates.dtype)[6:]}): {t:.1f} us  {fl / t / 1e6:.0f} TF/s", flush=True)
t = timeit(lambda: eng.step(T(x), h0, c0, n, H, W, h1, c1, None))
print(f"{tag:>20} lstm cell fwd (no gates saved): {t:.1f} us", flush=True)
