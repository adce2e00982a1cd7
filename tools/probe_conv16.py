"""Timing of the bf16-MFMA convolution at the MetNet encoder shapes with bf16-STORED activations ("bf16a")."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, satflow_amd
from satflow_amd import kernels as K
from satflow_amd._hip import T, NULL, cpad
from satflow_amd.functional import ConvEngine
satflow_amd.set_compute_dtype("bf16a")
dev = torch.device("cuda:0")
tag = os.environ.get("SATFLOW_HIP_LIB", "default")[-22:]
for (n, cin, cout, H, W, stats) in ((2304, 256, 256, 32, 32, False), (2304, 256, 256, 32, 32, True), (2304, 160, 256, 32, 32, True), (2304, 256, 160, 32, 32, False), (192, 96, 160, 64, 64, False)):
    eng = ConvEngine([cin], cout)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.02; b = torch.randn(cout, device=dev)
    packed, bp = K.pack_weights(w, b, eng.fwd_map, False)
    x = torch.randn(n, H, W, cpad(cin), device=dev).bfloat16(); y = torch.empty(n, H, W, cpad(cout), device=dev, dtype=torch.bfloat16)
    st = torch.empty(n * int(satflow_amd._hip.lib().sf_conv3x3_stats_tiles(H, W)), eng.fwd_map.Np, 2, device=dev) if stats else None
    f = lambda: K.conv3x3(T(x), NULL, n, H, W, packed, bp, eng.fwd_map, T(y), stats=st)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{tag:>22} conv {cin}->{cout} {H}x{W} n={n} stats={int(stats)}: {ms:.3f} ms  {2*9*cin*cout*H*W*n/ms/1e9:.0f} TF/s", flush=True)
