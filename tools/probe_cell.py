"""Micro-benchmark of the fused cell / conv kernels at BASELINE cfg-2 shapes (GPU box only)."""
import sys, time
import torch
sys.path.insert(0, ".")
from satflow_amd import kernels as K
from satflow_amd._hip import T, NULL, cpad
from satflow_amd.models.layers.ConvLSTM import CellEngine
from torch import nn

dev = torch.device("cuda:0")

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

def cell(B, cin, hid, H, W):
    conv = nn.Conv2d(cin + hid, 4 * hid, 3, padding=1).to(dev)
    eng = CellEngine(conv, cin, hid)
    x = torch.randn(B, H, W, cpad(cin), device=dev)
    h = torch.randn(B, H, W, hid, device=dev); c = torch.randn(B, H, W, hid, device=dev)
    ho, co = torch.empty_like(h), torch.empty_like(c)
    g = torch.empty(B, H, W, 4 * hid, device=dev)
    eng.packed_fwd()
    t = timeit(lambda: eng.step(T(x), h, c, B, H, W, ho, co, g))
    flops = 2 * 9 * (cin + hid) * 4 * hid * H * W * B
    bytes_ = (cin + 2 * hid + 2 * hid) * H * W * B * 4
    print(f"cell fwd B={B} cin={cin} hid={hid} {H}x{W}: {t*1e6:8.1f} us  {flops/t/1e12:6.1f} TF/s  alg {bytes_/t/1e9:7.1f} GB/s")
    dz = torch.randn(B, H, W, 4 * hid, device=dev)
    dcat = torch.empty(B, H, W, cpad(cin) + hid, device=dev)
    eng.packed_bwd(True)
    t = timeit(lambda: eng.bwd_data(dz, B, H, W, True, dcat))
    print(f"   bwd-data: {t*1e6:8.1f} us  {2*9*4*hid*(cpad(cin)+hid)*H*W*B/t/1e12:6.1f} TF/s")
    dw = torch.empty_like(conv.weight); db = torch.empty_like(conv.bias)
    t = timeit(lambda: eng.bwd_weight(T(x), T(h), T(dz), B, H, W, dw, db, False), iters=5)
    print(f"   bwd-weight: {t*1e6:8.1f} us  {flops/t/1e12:6.1f} TF/s")
    dh = torch.randn_like(h); dc = torch.randn_like(c); dzo = torch.empty_like(dz)
    t = timeit(lambda: eng.bwd_gates([T(dh)], dc, g, c, co, dzo, dc))
    print(f"   bwd-gates: {t*1e6:8.1f} us  {(hid*(1+1+4+1+1+4+1))*H*W*B*4/t/1e9:7.1f} GB/s")

for B in (1, 8):
    cell(B, 12, 64, 128, 128)
    cell(B, 64, 64, 128, 128)
