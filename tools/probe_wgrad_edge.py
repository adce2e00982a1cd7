"""Timing probe: sf_conv3x3_bwd_weight on bf16-stored tensors at the MetNet shapes with half-empty edge tiles (SF_NO_WGRAD_WIDE=1: regular slabs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, satflow_amd, bench
from satflow_amd import kernels as K
from satflow_amd._hip import T, cpad
from satflow_amd.functional import ConvEngine
satflow_amd.set_compute_dtype("bf16a")
dev = torch.device("cuda:0")
for cin, cout, n, h, w in ((256, 192, 2304, 16, 16), (108, 160, 192, 64, 64), (160, 256, 2304, 32, 32), (256, 256, 2304, 32, 32)):
    eng = ConvEngine([cin], cout)
    x = torch.randn(n, h, w, cpad(cin), device=dev).to(torch.bfloat16)
    gy = torch.randn(n, h, w, eng.coutp, device=dev).to(torch.bfloat16)
    dw, db = torch.empty(cout, cin, 3, 3, device=dev), torch.empty(cout, device=dev)
    t = bench.event_time(lambda: K.conv3x3_bwd_weight(T(x), T(None), T(gy), n, h, w, eng.wgrad_map, dw, db, False), iters=10)
    fl = 2 * 9 * cin * cout * h * w * n
    print(f"wgrad {cin}->{cout} @{h}x{w} x {n}: {t*1e3:.3f} ms = {fl/t/1e12:.0f} TF/s ({os.environ.get('SF_NO_WGRAD_WIDE') and 'regular slabs' or 'edge slabs'})")
