"""Timing probe: the ConvGRU-side convolutions of MetNet cfg3 (2304 images of 16x16) in bf16 kernels, fp32 storage."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, satflow_amd
from satflow_amd import kernels as K
from satflow_amd._hip import T, NULL
from satflow_amd.functional import ConvEngine
satflow_amd.set_compute_dtype("bf16")
dev = torch.device("cuda:0")
for (n, cin, cout, H, W) in ((2304, 256, 192, 16, 16), (2304, 192, 256, 16, 16), (96, 64, 192, 16, 16), (96, 192, 64, 16, 16), (2305, 64, 64, 8, 8)):
    eng = ConvEngine([cin], cout)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.02; b = torch.randn(cout, device=dev)
    packed, bp = K.pack_weights(w, b, eng.fwd_map, False)
    x = torch.randn(n, H, W, eng.fwd_map.Kp, device=dev); y = torch.empty(n, H, W, eng.coutp, device=dev)
    f = lambda: K.conv3x3(T(x), NULL, n, H, W, packed, bp, eng.fwd_map, T(y))
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    fl = 2 * 9 * cin * cout * H * W * n
    print(f"{os.environ.get('SATFLOW_HIP_LIB','default')[-22:]:>22} conv {cin}->{cout} {H}x{W} n={n} nf={eng.fwd_map.nf}: {ms*1e3:.1f} us  {fl/ms/1e9:.0f} TF/s", flush=True)
