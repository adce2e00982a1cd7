"""Does the time of an f32e convolution depend on the gradient operand's scale word?  conv 256 -> 256 @32x32 x 2304 (MetNet's conv3 / conv4 shape), input-gradient form,
the same random operand with the amax word set to its true maximum and to wrong values; forward form (no word) and the bf16 kernel beside it.   (GPU)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, satflow_amd
from satflow_amd import kernels as K
from satflow_amd._hip import T, NULL
from satflow_amd.functional import ConvEngine

dev = torch.device("cuda:0")
n, C, H = int(os.environ.get("SF_N", 2304)), 256, 32
eng = ConvEngine([C], C)
w = torch.randn(C, C, 3, 3, device=dev) * 0.02
gscale = float(os.environ.get("SF_GSCALE", 1.0))
x = torch.randn(n, H, H, C, device=dev) * gscale
y = torch.empty(n, H, H, C, device=dev)


def timeit(f, iters=5):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for mode in ("bf16", "f32e"):
    satflow_amd.set_compute_dtype(mode)
    packed, _ = K.pack_weights(w, None, eng.fwd_map, False)
    t = timeit(lambda: K.conv3x3(T(x), NULL, n, H, H, packed, None, eng.fwd_map, T(y)))
    print(f"{mode:5s} no scale word                      {t:7.3f} ms", flush=True)
    if mode == "f32e":
        true = float(x.abs().max())
        for tag, val in (("true amax", true), ("amax / 2", true / 2), ("amax / 4", true / 4), ("amax x 16", true * 16), ("amax x 4096", true * 4096)):
            word = torch.full((1,), val, device=dev)
            t = timeit(lambda: K.conv3x3(T(x, amax=word), NULL, n, H, H, packed, None, eng.fwd_map, T(y)))
            print(f"f32e  word = {tag:12s} ({val:10.3e})   {t:7.3f} ms   finite {bool(torch.isfinite(y).all())}", flush=True)
