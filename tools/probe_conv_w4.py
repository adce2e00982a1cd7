"""A/B of the one-wave-per-SIMD persistent convolution (conv3x3_bf16_persist4.hip) against the 8-wave persistent kernel: run once
as is and once with SF_NO_CONV_W4=1 in the same gpurun call.  Shapes: MetNet's folded 256 -> 256 and 160 -> 256 @32x32 x 2304."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import satflow_amd  # noqa: E402
from satflow_amd import kernels as K  # noqa: E402
from satflow_amd._hip import T, cpad  # noqa: E402
from satflow_amd.functional import ConvEngine  # noqa: E402

dev = torch.device("cuda:0")
satflow_amd.set_compute_dtype("bf16a")


def timeit(fn, iters=20, warm=5):
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


for cin, cout in ((256, 256), (160, 256)):
    n, H, W, groups = 2304, 32, 32, 12
    eng = ConvEngine([cin], cout)
    gm = eng.fwd_map
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.03
    b = torch.randn(cout, device=dev)
    scale = 0.5 + torch.rand(groups, gm.Kp, device=dev)
    shift = torch.randn(groups, gm.Kp, device=dev)
    packed, tab = K.conv3x3_fold_pack(w, b, gm, scale, shift)
    x = torch.randn(n, H, W, cpad(cin), device=dev).to(torch.bfloat16)
    y = torch.empty(n, H, W, eng.coutp, device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: K.conv3x3_folded(T(x), n, H, W, packed, tab, gm, T(y)))
    fl = 2 * 9 * cin * cout * H * W * n
    tag = f"W4 {'off' if os.environ.get('SF_NO_CONV_W4') else 'on'}{', statistics launches row-major' if os.environ.get('SF_CONV_W4_WIN') == '0' else ''}"
    print(f"folded conv {cin}->{cout} @32x32 x {n} ({tag}): {ms:.3f} ms = {fl / ms / 1e9:.0f} TF/s = {fl / ms / 1e9 / 2500:.3f} of 2.5 PF")
    from satflow_amd._hip import lib  # noqa: E402
    st = torch.empty(n * int(lib().sf_conv3x3_stats_tiles(H, W)), gm.Np, 2, device=dev)
    ms = timeit(lambda: K.conv3x3_folded(T(x), n, H, W, packed, tab, gm, T(y), stats=st))
    print(f"   + BatchNorm statistics ({tag}): {ms:.3f} ms = {fl / ms / 1e9 / 2500:.3f} of 2.5 PF")
    if K.conv3x3_folded_pool_supported(n, H, W, gm, eng.coutp, groups):
        ms = timeit(lambda: K.conv3x3_folded_pool(T(x), n, H, W, packed, tab, gm, eng.coutp, (12, 24), None, dev))
        print(f"   + 2x2 max-pooling in the epilogue (incl. the two output allocations): {ms:.3f} ms = {fl / ms / 1e9 / 2500:.3f} of 2.5 PF")

# the input gradient with the BatchNorm-backward epilogue (sf_conv3x3_bwd_data_bn; SF_NO_CONV_W4_BNB=1: the 8-wave one-item kernel)
for cin, cout in ((256, 256),):
    n, H, W, groups = 2304, 32, 32, 12
    eng = ConvEngine([cin], cout)
    gmb = eng.bwd_map((True,))
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.03
    packed_t = eng.packed(w, None, "bwd", (True,))[0]
    gy = torch.randn(n, H, W, eng.coutp, device=dev).to(torch.bfloat16)
    x = torch.randn(n, H, W, cpad(cin), device=dev).to(torch.bfloat16)
    coef = torch.randn(groups, 3, cpad(cin), device=dev)
    dx = torch.empty_like(x)
    ms = timeit(lambda: K.conv3x3_bwd_data_bn(T(gy), n, H, W, packed_t, gmb, T(x), coef, T(dx)))
    fl = 2 * 9 * cin * cout * H * W * n
    print(f"input gradient + BatchNorm backward {cout}->{cin} @32x32 x {n} (W4 {'off' if os.environ.get('SF_NO_CONV_W4_BNB') else 'on'}): {ms:.3f} ms = {fl / ms / 1e9:.0f} TF/s = {fl / ms / 1e9 / 2500:.3f} of 2.5 PF")
