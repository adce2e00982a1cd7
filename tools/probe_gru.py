"""Timing probe of the ConvGRU recurrent step (MetNet cfg3: 96 images 16x16, hidden 64) and its backward conv."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, satflow_amd
from satflow_amd import kernels as K
from satflow_amd._hip import T, NULL
from satflow_amd.functional import GRUEngine
satflow_amd.set_compute_dtype(os.environ.get("SF_MODE", "bf16"))
dev = torch.device("cuda:0")
n, H, W, cin, hid = 96, 16, 16, 256, 64
eng = GRUEngine(cin, hid)
Wx = torch.randn(3 * hid, cin, 3, 3, device=dev) * 0.02; bx = torch.zeros(3 * hid, device=dev)
Wh = torch.randn(3 * hid, hid, 3, 3, device=dev) * 0.05; bh = torch.zeros(3 * hid, device=dev)
pk = eng.packed(Wx, bx, Wh, bh)
gx = torch.randn(n, H, W, 3 * eng.hidp, device=dev)
h0 = torch.randn(n, H, W, eng.hidp, device=dev); h1 = torch.empty_like(h0)
gates = torch.empty(n, H, W, 4 * eng.hidp, device=dev)
dgh = torch.randn(n, H, W, 3 * eng.hidp, device=dev); carry = torch.empty_like(h0)
def timeit(f, iters=48):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
t = timeit(lambda: K.convgru_step_fwd(T(gx), h0, n, H, W, pk["h_fwd"][0], pk["h_fwd"][1], eng.hidp, h1, gates))
print(f"gru step fwd: {t:.1f} us")
t = timeit(lambda: K.conv3x3(T(dgh), NULL, n, H, W, pk["h_bwd"], None, eng.h_bwd, T(carry)))
print(f"gru step bwd conv (192->64): {t:.1f} us")
t = timeit(lambda: K.convgru_bwd_gates([T(h1)], gates, h0, eng.hidp, gx, dgh, carry))
print(f"gru bwd gates: {t:.1f} us")
