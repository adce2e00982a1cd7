"""Timing probe of sf_conv3x3_bwd_weight at the MetNet shapes.  SF_ACT=bf16 stores the tensors as bf16 ("bf16a" mode)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, satflow_amd
from satflow_amd import kernels as K
from satflow_amd._hip import T, NULL
from satflow_amd.functional import ConvEngine
satflow_amd.set_compute_dtype("bf16")
dev = torch.device("cuda:0")
st = torch.bfloat16 if os.environ.get("SF_ACT", "f32") == "bf16" else torch.float32
if os.environ.get("SF_SHAPES") == "gru":   # the generator's half-resolution ConvGRU weight gradients (DGMR line)
    shapes = ((14, 2048, 4096, 32, 32), (14, 1024, 2048, 64, 64), (16, 1024, 4096, 32, 32), (14, 2048, 4096, 16, 16), (16, 512, 2048, 64, 64),
              (14, 2048, 4096, 8, 8), (16, 512, 512, 64, 64), (16, 128, 128, 256, 256), (16, 256, 256, 128, 128))
else:
  shapes = ((2304, 256, 256, 32, 32), (2304, 160, 256, 32, 32), (192, 96, 160, 64, 64), (2304, 256, 192, 16, 16))
for (n, cin, cout, H, W) in shapes[: int(os.environ.get('SF_ONLY', len(shapes)))]:
    eng = ConvEngine([cin], cout)
    x = torch.randn(n, H, W, eng.fwd_map.Kp, device=dev).to(st); gy = torch.randn(n, H, W, eng.coutp, device=dev).to(st)
    dw = torch.empty(cout, cin, 3, 3, device=dev); db = torch.empty(cout, device=dev)
    f = lambda: K.conv3x3_bwd_weight(T(x), NULL, T(gy), n, H, W, eng.wgrad_map, dw, db, False)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 2 * 9 * cin * cout * H * W * n
    print(f"{os.environ.get('SATFLOW_HIP_LIB','default')[-24:]:>24} act={st} wgrad {cin}->{cout} {H}x{W} n={n}: {ms:.3f} ms  {fl/ms/1e9:.0f} TF/s", flush=True)
