"""Per-shape census of the DGMR step's big kernels: every conv3x3 / weight-gradient / linear / bmm call of ONE eager step, timed with HIP events,
grouped by shape.  Prints calls, total ms, FLOPs and achieved TF/s per group - the work list for the DGMR kernels."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import satflow_amd
from satflow_amd import kernels as K
import bench

satflow_amd.set_compute_dtype(sys.argv[1] if len(sys.argv) > 1 else "bf16a")
os.environ["SF_NO_GRAPH"] = "1"
dev = torch.device("cuda:0")
wl = bench.DGMRWorkload(dev, 2, 0)
for _ in range(2):
    wl.step()
torch.cuda.synchronize()
log = []


def timed(name, fn, describe):
    def wrapper(*a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **kw)
        e1.record()
        log.append((name, describe(*a, **kw), e0, e1))
        return r
    return wrapper


def d_conv(s0, s1, n, h, w, packed, bp, gm, out, *a, **kw):
    return (n, h, w, gm.Kp, gm.Np), 2.0 * 9 * n * h * w * gm.Kp * gm.Np


def d_wgrad(s0, s1, dout, n, h, w, gm, dw, db, accumulate=False):
    return (n, h, w, s0.c + s1.c, dout.c), 2.0 * 9 * n * h * w * (s0.c + s1.c) * dout.c


def d_lin(x, W, bias, out_lanes, lowp=False):
    rows = x.numel() // x.shape[-1]
    return (rows, x.shape[-1], W.shape[0]), 2.0 * rows * x.shape[-1] * W.shape[0]


def d_bmm(A, B, out, alpha=1.0, beta=0.0, lowp=False):
    return (tuple(A.shape), tuple(B.shape)), 2.0 * A.shape[0] * A.shape[1] * A.shape[2] * B.shape[2]


def d_conv5(x, n, h, w, packed, bp, gm, out):   # shifted-view 5x5 route: K = four views; 25 of the 36 taps are multiplied
    return (n, h, w, gm.Kp, gm.Np, "5x5 shifted views"), 2.0 * 25 / 4 * n * h * w * gm.Kp * gm.Np


def d_wgrad5(x, dout, n, h, w, gm, dw, db):
    return (n, h, w, 4 * x.shape[-1], dout.shape[-1], "5x5 shifted views"), 2.0 * 25 / 4 * n * h * w * 4 * x.shape[-1] * dout.shape[-1]


K.conv5x5_shift4 = timed("conv5x5", K.conv5x5_shift4, d_conv5)
K.conv5x5_shift4_bwd_weight = timed("wgrad5x5", K.conv5x5_shift4_bwd_weight, d_wgrad5)
K.conv3x3 = timed("conv3x3", K.conv3x3, d_conv)
K.conv3x3_bwd_weight = timed("wgrad", K.conv3x3_bwd_weight, d_wgrad)
K.linear_fwd = timed("linear_fwd", K.linear_fwd, d_lin)
K.bmm_raw = timed("bmm", K.bmm_raw, d_bmm)
from satflow_amd import functional_gan as FG
FG._bmm_raw = K.bmm_raw
wl.step()
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for name, (shape, flops), e0, e1 in log:
    a = agg[(name, shape)]
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
    a[2] += flops
tot = collections.defaultdict(float)
for (name, shape), (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot[name] += ms
    print(f"{name:10s} {str(shape):60s} calls {n:4d}  {ms:8.2f} ms  {fl / 1e9:9.1f} GF  {fl / ms / 1e9 if ms else 0:8.1f} TF/s")
print({k: round(v, 1) for k, v in tot.items()})
