"""Timing probe: the x-part convolution of MetNet's ConvGRU (256 -> 192 lanes on 2304 maps of 16x16, bf16-stored input and output) and its input
gradient (192 -> 256); SF_CONV_NO_DUAL_NF3=1: the NF = 3 forward on the 4-wave 16x16 kernel instead of the two-image 8-wave kernel (the default since round 5)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, satflow_amd
from satflow_amd import kernels as K
from satflow_amd._hip import T, NULL
from satflow_amd.functional import GRUEngine
satflow_amd.set_compute_dtype("bf16a")
dev = torch.device("cuda:0")
n, H, W, cin, hid = 2304, 16, 16, 256, 64
eng = GRUEngine(cin, hid)
Wx = torch.randn(3 * hid, cin, 3, 3, device=dev) * 0.02; bx = torch.zeros(3 * hid, device=dev)
Wh = torch.randn(3 * hid, hid, 3, 3, device=dev) * 0.05; bh = torch.zeros(3 * hid, device=dev)
pk = eng.packed(Wx, bx, Wh, bh)
x = torch.randn(n, H, W, cin, device=dev).bfloat16()
gx = torch.empty(n, H, W, 3 * eng.hidp, device=dev, dtype=torch.bfloat16)
dgx = torch.randn(n, H, W, 3 * eng.hidp, device=dev).bfloat16()
dx = torch.empty_like(x)
def timeit(f, iters=20):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
fl = 2 * 9 * cin * 3 * hid * H * W * n
t = timeit(lambda: K.conv3x3(T(x), NULL, n, H, W, pk["x_fwd"][0], pk["x_fwd"][1], eng.x_fwd, T(gx)))
print(f"x-part forward 256->192 (nf={eng.x_fwd.nf}, two-image kernel {'off' if os.environ.get('SF_CONV_NO_DUAL_NF3') else 'on'}): {t:.1f} us = {fl / t / 1e6:.0f} TF/s")
t = timeit(lambda: K.conv3x3(T(dgx), NULL, n, H, W, pk["x_bwd"], None, eng.x_bwd, T(dx)))
print(f"x-part input gradient 192->256 (nf={eng.x_bwd.nf}): {t:.1f} us = {fl / t / 1e6:.0f} TF/s")
