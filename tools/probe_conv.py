"""Micro-benchmark of the conv kernels (fp32 / bf16 modes) at MetNet and ConvLSTM shapes (GPU box only)."""
import sys
import torch
sys.path.insert(0, ".")
import satflow_amd
from satflow_amd import kernels as K
from satflow_amd._hip import T, NULL, cpad
from satflow_amd.functional import ConvEngine
from satflow_amd.models.layers.ConvLSTM import CellEngine
from torch import nn

dev = torch.device("cuda:0")

def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

def conv(n, cin, cout, H, W):
    eng = ConvEngine([cin], cout)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.02; b = torch.randn(cout, device=dev)
    packed, bp = K.pack_weights(w, b, eng.fwd_map, False)
    x = torch.randn(n, H, W, cpad(cin), device=dev); y = torch.empty(n, H, W, cpad(cout), device=dev)
    t = timeit(lambda: K.conv3x3(T(x), NULL, n, H, W, packed, bp, eng.fwd_map, T(y)))
    fl = 2 * 9 * cin * cout * H * W * n
    print(f"  conv {cin}->{cout} {H}x{W} n={n} nf={eng.fwd_map.nf}: {t*1e3:8.3f} ms  {fl/t/1e12:7.1f} TF/s  alg {(cin+cout)*H*W*n*4/t/1e9:7.0f} GB/s")

def wgrad(n, cin, cout, H, W):
    eng = ConvEngine([cin], cout)
    x = torch.randn(n, H, W, cpad(cin), device=dev); gy = torch.randn(n, H, W, cpad(cout), device=dev)
    dw = torch.empty(cout, cin, 3, 3, device=dev); db = torch.empty(cout, device=dev)
    t = timeit(lambda: K.conv3x3_bwd_weight(T(x), NULL, T(gy), n, H, W, eng.wgrad_map, dw, db, False), iters=5)
    fl = 2 * 9 * cin * cout * H * W * n
    print(f"  wgrad {cin}->{cout} {H}x{W} n={n}: {t*1e3:8.3f} ms  {fl/t/1e12:7.1f} TF/s")

def cell(B, cin, hid, H, W):
    c = nn.Conv2d(cin + hid, 4 * hid, 3, padding=1).to(dev)
    eng = CellEngine(c, cin, hid)
    x = torch.randn(B, H, W, cpad(cin), device=dev)
    h = torch.randn(B, H, W, hid, device=dev); cc = torch.randn(B, H, W, hid, device=dev)
    ho, co = torch.empty_like(h), torch.empty_like(cc); g = torch.empty(B, H, W, 4 * hid, device=dev)
    eng.packed_fwd()
    t = timeit(lambda: eng.step(T(x), h, cc, B, H, W, ho, co, g))
    fl = 2 * 9 * (cin + hid) * 4 * hid * H * W * B
    print(f"  lstm cell {cin}+{hid} {H}x{W} B={B}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s")

for mode in ("f32", "bf16"):
    satflow_amd.set_compute_dtype(mode)
    print(mode)
    conv(2304, 256, 256, 32, 32)
    conv(2304, 160, 256, 32, 32)
    conv(2304, 256, 160, 32, 32)
    conv(2304, 112, 160, 64, 64)
    conv(2304, 256, 192, 16, 16)
    wgrad(2304, 256, 256, 32, 32)
    wgrad(2304, 160, 256, 32, 32)
    wgrad(2304, 112, 160, 64, 64)
    wgrad(96 * 12, 128, 256, 128, 128) if False else None
    cell(8, 64, 64, 128, 128)
    cell(8, 12, 64, 128, 128)
