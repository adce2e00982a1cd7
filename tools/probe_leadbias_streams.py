"""How does the lead-time pooling scale with the number of lead times (= concurrent output / gradient streams)?  GB/s per L."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import satflow_amd  # noqa: F401
from satflow_amd.functional import leadtime_pool

dev = torch.device("cuda:0")
Fr, H, C = 192, 64, 160


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for L in (1, 2, 3, 4, 6, 8, 12):
    base = torch.randn(Fr, H, H, C, device=dev).to(torch.bfloat16).requires_grad_()
    w = (torch.randn(160, 96 + L, 3, 3, device=dev) * 0.1).requires_grad_()
    y, _ = leadtime_pool(base, w, 96, L, want_stats=True)
    g = torch.randn_like(y)

    def bwd():
        base.grad = None
        w.grad = None
        y.backward(g, retain_graph=True)

    tf = timeit(lambda: leadtime_pool(base.detach(), w.detach(), 96, L, want_stats=True))
    tb = timeit(bwd)
    rb = Fr * H * H * C * 2
    ob = L * Fr * (H // 2) ** 2 * C * 2
    print(f"L={L:2d}: fwd {tf:.3f} ms = {(rb + ob) / tf / 1e9:6.0f} GB/s   bwd (all kernels) {tb:.3f} ms = {(2 * rb + ob) / tb / 1e9:6.0f} GB/s")
