"""BatchNorm -> Conv2d fold (sf_conv3x3_fold_pack / sf_conv3x3_fwd_folded / sf_conv3x3_bwd_weight_folded; the DownSampler pairs of
metnet.MetNet, call site satflow/models/pl_metnet.py:46-59).

The kernels are checked against float64 evaluations of EXACTLY the products they form - bf16-stored x and dout, weights
``bf16(W * scale_g)``, fp32 scale / shift as the BatchNorm kernel wrote them - so the bound is fp32 accumulation noise, not
bf16 rounding; then the folded DownSampler against the unfolded one (two roundings of the same quantity: bf16 tolerance)."""
import os

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _restore_compute_dtype():
    import satflow_amd

    yield
    satflow_amd.set_compute_dtype("f32")


def _setup(device, n, groups, cin, cout, H, W, seed=0):
    import satflow_amd
    from satflow_amd import functional as F

    satflow_amd.set_compute_dtype("bf16")
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(n, H, W, cin, generator=g) * 1.7 + 0.4).to(device).to(torch.bfloat16)
    bn = torch.nn.BatchNorm2d(cin).to(device)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(cin, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(cin, generator=g) * 0.3)
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1).to(device)
    eng = F.ConvEngine([cin], cout)
    return F, x, bn, conv, eng


def _ref_stats(x, bn, groups):
    """scale / shift per group from the bf16-stored x (biased variance), rounded where bn_finalize_kernel rounds: rstd to fp32,
    scale = gamma * rstd in fp32 (the packed weights are bf16(W * scale): a different last bit of scale would flip roundings)."""
    n, H, W, C = x.shape
    xg = x.double().reshape(groups, -1, C)
    mean = xg.mean(1)
    var = xg.var(1, unbiased=False)
    a = (bn.weight.float() * (1.0 / torch.sqrt(var + bn.eps)).float()).double()
    return a, bn.bias.double() - mean * a


@pytest.mark.parametrize("n,groups,cin,cout,H,W,want_stats", [
    (6, 3, 32, 64, 20, 20, True),      # ragged 32x16 tiles, statistics epilogue
    (6, 3, 32, 64, 20, 20, False),     # transposed epilogue
    (4, 2, 160, 256, 32, 32, True),    # DownSampler conv 2 (NF = 4 stats kernel)
    (4, 1, 256, 256, 32, 32, False),   # one group
    (8, 4, 48, 160, 12, 10, True),     # 4-wave kernel, NF = 5
    (4, 2, 16, 32, 2, 2, False),       # every pixel a corner
    (520, 4, 32, 64, 32, 32, True),    # 1040 tiles: the persistent kernel (one workgroup per CU walks several items), statistics epilogue
    (516, 4, 32, 160, 40, 24, False),  # large launch with NF = 5: stays on the one-item kernel
    (528, 4, 48, 128, 36, 20, False),  # persistent kernel, transposed epilogue, ragged tiles
])
def test_folded_forward_matches_float64_of_its_operands(device, n, groups, cin, cout, H, W, want_stats):
    F, x, bn, conv, eng = _setup(device, n, groups, cin, cout, H, W)
    bn.train()
    y, st = F.batchnorm_conv3x3(x, bn, groups, None, eng, conv.weight, conv.bias, out_dtype=torch.float32, want_stats=want_stats)
    assert (st is not None) == want_stats
    a, b = _ref_stats(x, bn, groups)
    ipg = n // groups
    for g in range(groups):
        wg = (conv.weight.float() * a[g].float().view(1, -1, 1, 1)).to(torch.bfloat16).double()  # what fold_pack stores
        xs = x[g * ipg:(g + 1) * ipg].double().permute(0, 3, 1, 2)
        ref = torch.nn.functional.conv2d(xs, wg, None, padding=1)
        shift_img = (b[g] / a[g]).view(1, -1, 1, 1).expand(1, cin, H, W)  # conv(a x + b) = conv_{W a}(x + b / a), same rounded weights
        ref = ref + torch.nn.functional.conv2d(shift_img, wg, conv.bias.double(), padding=1)
        got = y[g * ipg:(g + 1) * ipg, ..., :cout].permute(0, 3, 1, 2).double()
        assert rel_l2(got, ref) < 1e-4, f"group {g}: rel L2 {rel_l2(got, ref):.3e}"
    if want_stats:  # the emitted statistics describe the stored tensor
        tiles = st.tiles
        s = st.data.reshape(n, tiles, st.np, 2).double().sum(1)[:, :cout]
        assert rel_l2(s[..., 0], y[..., :cout].double().sum((1, 2))) < 1e-5
        assert rel_l2(s[..., 1], (y[..., :cout].double() ** 2).sum((1, 2))) < 1e-5


def test_folded_forward_with_vanishing_gamma(device):
    """A pruned / decayed BatchNorm channel: gamma * rstd tiny but nonzero.  shift / scale overflows while bf16(w * scale) flushes to
    zero - the table must take the exact w * shift branch (inf * 0 would put NaN into every pixel of the group) and agree with the
    unfolded BatchNorm -> convolution."""
    n, groups, cin, cout, H, W = 4, 2, 32, 64, 12, 12
    F, x, bn, conv, eng = _setup(device, n, groups, cin, cout, H, W)
    with torch.no_grad():
        bn.weight[3] = 1e-30
        bn.weight[7] = 0.0
        bn.weight[11] = -1e-38
        bn.bias[3], bn.bias[7], bn.bias[11] = 0.7, -0.4, 1.3
    bn.train()
    y, _ = F.batchnorm_conv3x3(x, bn, groups, None, eng, conv.weight, conv.bias, out_dtype=torch.float32, want_stats=False)
    assert torch.isfinite(y).all()
    a, b = _ref_stats(x, bn, groups)
    ipg = n // groups
    for g in range(groups):
        xs = x[g * ipg:(g + 1) * ipg].double().permute(0, 3, 1, 2)
        xn = xs * a[g].view(1, -1, 1, 1) + b[g].view(1, -1, 1, 1)
        ref = torch.nn.functional.conv2d(xn, conv.weight.double(), conv.bias.double(), padding=1)
        got = y[g * ipg:(g + 1) * ipg, ..., :cout].permute(0, 3, 1, 2).double()
        assert rel_l2(got, ref) < 2e-2, f"group {g}: rel L2 {rel_l2(got, ref):.3e}"  # bf16 operands vs float64


@pytest.mark.parametrize("n,groups,cin,cout,H,W", [
    (6, 3, 32, 64, 20, 20),
    (12, 3, 160, 256, 32, 32),   # slices cross group boundaries (two segments per slice)
    (24, 24, 64, 128, 16, 16),   # slices longer than a group (several boundaries per slice)
    (4, 2, 16, 32, 2, 2),
])
def test_folded_weight_gradient_matches_float64_of_its_operands(device, n, groups, cin, cout, H, W):
    from satflow_amd import kernels as K
    from satflow_amd._hip import T

    F, x, bn, conv, eng = _setup(device, n, groups, cin, cout, H, W, seed=1)
    g = torch.Generator().manual_seed(5)
    gy = torch.randn(n, H, W, eng.coutp, generator=g).to(device).to(torch.bfloat16)
    gy[..., cout:] = 0
    a, b = _ref_stats(x, bn, groups)
    scale, shift = a.float().contiguous(), b.float().contiguous()
    dw = torch.empty(cout, cin, 3, 3, device=device)
    db = torch.empty(cout, device=device)
    K.conv3x3_bwd_weight_folded(T(x), T(gy), n, H, W, eng.wgrad_map, scale, shift, dw, db)
    ipg = n // groups
    w = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, device=device, requires_grad=True)
    tot = 0
    for gi in range(groups):
        xs = x[gi * ipg:(gi + 1) * ipg].double().permute(0, 3, 1, 2)
        xin = xs * scale[gi].double().view(1, -1, 1, 1) + shift[gi].double().view(1, -1, 1, 1)
        out = torch.nn.functional.conv2d(xin, w, None, padding=1)
        tot = tot + (out * gy[gi * ipg:(gi + 1) * ipg, ..., :cout].double().permute(0, 3, 1, 2)).sum()
    (ref,) = torch.autograd.grad(tot, w)
    assert rel_l2(dw, ref) < 2e-5, f"dW rel L2 {rel_l2(dw, ref):.3e}"
    assert rel_l2(db, gy[..., :cout].double().sum((0, 1, 2))) < 2e-5


def test_folded_downsampler_is_as_close_to_fp32_as_unfolded(device, monkeypatch):
    """Whole DownSampler (three folded pairs), forward + all gradients.  Folded and unfolded are two bf16 roundings of the same
    quantities; the backward through three BatchNorms amplifies rounding noise to several percent in EITHER form (the bf16a
    yardstick of test_bf16a_gpu.py), so each is measured against the fp32 kernels: the fold must not be the noisier one."""
    import satflow_amd
    from satflow_amd.models.metnet import DownSampler

    torch.manual_seed(3)
    ds = DownSampler(12, 256).to(device).train()
    x = torch.randn(12, 32, 32, 16, device=device)
    x[..., 12:] = 0
    gout = torch.randn(12, 8, 8, 256, device=device)

    def run(mode, fold):
        satflow_amd.set_compute_dtype("f32" if mode == "f32" else "bf16")
        if fold:
            monkeypatch.delenv("SF_NO_BN_FOLD", raising=False)
        else:
            monkeypatch.setenv("SF_NO_BN_FOLD", "1")
        for p in ds.parameters():
            p.grad = None
        for m in ds.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.reset_running_stats()
        xi = (x if mode == "f32" else x.to(torch.bfloat16)).clone().requires_grad_(True)
        y = ds.run(xi, 3, out_dtype=torch.float32)
        y.backward(gout)
        rs = {k: v.clone() for k, v in ds.state_dict().items() if "running" in k}
        return y.detach(), xi.grad.float(), {k: p.grad.clone() for k, p in ds.named_parameters()}, rs

    try:
        yr, dxr, gr, rr = run("f32", False)
        y1, dx1, g1, r1 = run("bf16", True)
        y0, dx0, g0, r0 = run("bf16", False)
    finally:
        satflow_amd.set_compute_dtype("bf16")

    def check(name, folded, unfolded, ref, floor):
        e1, e0 = rel_l2(folded, ref), rel_l2(unfolded, ref)
        assert e1 < max(1.5 * e0, floor), f"{name}: folded {e1:.3e} vs unfolded {e0:.3e} from fp32"

    check("y", y1, y0, yr, 5e-3)
    check("dx", dx1, dx0, dxr, 2e-2)
    for k in gr:
        if k.endswith("bias") and ("module.0" in k or "module.4" in k or "module.6" in k):
            continue  # conv biases in front of a BatchNorm: true gradient zero
        check(k, g1[k], g0[k], gr[k], 2e-2)
    for k in rr:
        check(k, r1[k].double(), r0[k].double(), rr[k].double(), 1e-3)


def test_folded_path_is_taken_in_bf16a_mode(device, monkeypatch):
    """Guard against a silent fallback: the DownSampler's training step must call the folded entry points."""
    import satflow_amd
    from satflow_amd import kernels as K
    from satflow_amd.models.metnet import DownSampler

    satflow_amd.set_compute_dtype("bf16")
    monkeypatch.delenv("SF_NO_BN_FOLD", raising=False)
    calls = []
    orig = K.conv3x3_folded
    monkeypatch.setattr(K, "conv3x3_folded", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    ds = DownSampler(12, 64).to(device).train()
    x = torch.randn(4, 16, 16, 16, device=device).to(torch.bfloat16)
    ds.run(x, 2)
    assert len(calls) == 3


@pytest.mark.parametrize("n,groups,cin,cout,H,W", [(6, 3, 32, 64, 20, 20), (12, 3, 160, 256, 32, 32), (8, 4, 64, 48, 16, 16)])
def test_bn_backward_sums_from_weight_gradient_partials(device, n, groups, cin, cout, H, W):
    """sum dn and sum dn * xhat (dn = conv^T(dout, W)) derived from V_g and the per-group raw weight gradient, against float64 sums
    over an explicitly formed dn."""
    from satflow_amd import kernels as K
    from satflow_amd._hip import T

    F, x, bn, conv, eng = _setup(device, n, groups, cin, cout, H, W, seed=7)
    g = torch.Generator().manual_seed(8)
    gy = torch.randn(n, H, W, eng.coutp, generator=g).to(device).to(torch.bfloat16)
    gy[..., cout:] = 0
    a, b = _ref_stats(x, bn, groups)
    xg = x.double().reshape(groups, -1, cin)
    mean, var = xg.mean(1), xg.var(1, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + bn.eps)
    dw = torch.empty(cout, cin, 3, 3, device=device)
    sums = torch.empty(groups, 2, cin, dtype=torch.float64, device=device)
    K.conv3x3_bwd_weight_folded(T(x), T(gy), n, H, W, eng.wgrad_map, a.float().contiguous(), b.float().contiguous(), dw, None,
                                bn=(conv.weight, mean.float().contiguous(), rstd.float().contiguous(), sums))
    dn = torch.nn.functional.conv_transpose2d(gy[..., :cout].double().permute(0, 3, 1, 2), conv.weight.double(), padding=1)  # [n, cin, H, W]
    dng = dn.permute(0, 2, 3, 1).reshape(groups, -1, cin)
    xhat = (xg - mean.float().double()[:, None]) * rstd.float().double()[:, None]
    assert rel_l2(sums[:, 0], dng.sum(1)) < 1e-4
    assert rel_l2(sums[:, 1], (dng * xhat).sum(1)) < 1e-4


@pytest.mark.parametrize("n,groups,cin,cout,H,W", [(6, 3, 32, 64, 20, 20), (4, 2, 160, 256, 32, 32), (8, 4, 48, 96, 12, 10)])
def test_input_gradient_with_batchnorm_backward_epilogue(device, n, groups, cin, cout, H, W):
    """sf_conv3x3_bwd_data_bn: dx = A * conv^T(dout, W) + B * x + K per (group, channel), float64 of the same operands."""
    from satflow_amd import kernels as K
    from satflow_amd._hip import T

    F, x, bn, conv, eng = _setup(device, n, groups, cin, cout, H, W, seed=9)
    g = torch.Generator().manual_seed(10)
    gy = torch.randn(n, H, W, eng.coutp, generator=g).to(device).to(torch.bfloat16)
    gy[..., cout:] = 0
    coef = torch.randn(groups, 3, cin, generator=g).to(device).contiguous()
    dx = torch.empty_like(x)
    packed_t = eng.packed(conv.weight, conv.bias, "bwd", (True,))[0]
    K.conv3x3_bwd_data_bn(T(gy), n, H, W, packed_t, eng.bwd_map((True,)), T(x), coef, T(dx))
    wb = conv.weight.to(torch.bfloat16).double()
    dn = torch.nn.functional.conv_transpose2d(gy[..., :cout].double().permute(0, 3, 1, 2), wb, padding=1).permute(0, 2, 3, 1)
    ipg = n // groups
    cg = coef.double().repeat_interleave(ipg, 0)  # [n, 3, cin]
    ref = cg[:, 0, None, None, :] * dn + cg[:, 1, None, None, :] * x.double() + cg[:, 2, None, None, :]
    assert rel_l2(dx.double(), ref) < 3e-3  # bf16 storage of the result
    assert rel_l2(dx.double(), ref.float().to(torch.bfloat16).double()) < 3e-4  # against the rounded reference: a few last-bit flips


def test_fused_batchnorm_backward_matches_the_three_kernel_form(device, monkeypatch):
    import satflow_amd
    from satflow_amd.models.metnet import DownSampler

    satflow_amd.set_compute_dtype("bf16")
    monkeypatch.delenv("SF_NO_BN_FOLD", raising=False)
    torch.manual_seed(4)
    ds = DownSampler(12, 256).to(device).train()
    x = torch.randn(12, 32, 32, 16, device=device).to(torch.bfloat16)
    gout = torch.randn(12, 8, 8, 256, device=device)

    def run(passes):
        if passes:
            monkeypatch.setenv("SF_BN_BWD_PASSES", "1")
        else:
            monkeypatch.delenv("SF_BN_BWD_PASSES", raising=False)
        for p in ds.parameters():
            p.grad = None
        xi = x.clone().requires_grad_(True)
        ds.run(xi, 3, out_dtype=torch.float32).backward(gout)
        return xi.grad.float(), {k: p.grad.clone() for k, p in ds.named_parameters()}

    dx1, g1 = run(False)
    dx0, g0 = run(True)
    # the three-kernel form rounds the gradient entering each BatchNorm to bf16 before reducing it, the fused form does not:
    # differences are that rounding, amplified by the two BatchNorms behind it
    assert rel_l2(dx1, dx0) < 3e-2
    for k in g0:
        if k.endswith("bias") and ("module.0" in k or "module.4" in k or "module.6" in k):
            continue
        assert rel_l2(g1[k], g0[k]) < 3e-2, f"{k}: {rel_l2(g1[k], g0[k]):.3e}"


@pytest.mark.parametrize("Fr,H,W,C,L,dtype", [(6, 16, 16, 160, 12, torch.bfloat16), (5, 12, 20, 32, 3, torch.float32), (2, 2, 2, 16, 1, torch.bfloat16)])
def test_leadtime_pool_statistics(device, Fr, H, W, C, L, dtype):
    """sf_leadtime_pool_fwd_stats: same pooled tensor as sf_leadtime_pool_fwd, plus per-lead-time sum / sum of squares of the stored values."""
    from satflow_amd import functional as F

    g = torch.Generator().manual_seed(11)
    cimg = 20
    base = torch.randn(Fr, H, W, C, generator=g).to(device).to(dtype)
    w1 = (torch.randn(C, cimg + L, 3, 3, generator=g) * 0.3).to(device)
    ref = F.leadtime_pool(base, w1, cimg, L)
    out, st = F.leadtime_pool(base, w1, cimg, L, want_stats=True)
    assert st is not None and torch.equal(out, ref)
    s = st.data.reshape(L, st.tiles, C, 2).double().sum(1)
    o = out.double().reshape(L, -1, C)
    assert rel_l2(s[..., 0], o.sum(1)) < 1e-5
    assert rel_l2(s[..., 1], (o * o).sum(1)) < 1e-5


def test_leadtime_pool_statistics_fall_back_beyond_the_register_budget(device):
    """More than 12 lead times: no statistics records (the BatchNorm then reduces the pooled tensor itself), same pooled tensor."""
    from satflow_amd import functional as F

    g = torch.Generator().manual_seed(12)
    base = torch.randn(2, 8, 8, 32, generator=g).to(device)
    w1 = (torch.randn(32, 4 + 13, 3, 3, generator=g) * 0.3).to(device)
    out, st = F.leadtime_pool(base, w1, 4, 13, want_stats=True)
    assert st is None and torch.equal(out, F.leadtime_pool(base, w1, 4, 13))


def xn64(x, scale, shift, cin, groups):
    """float64 NCHW of what a convolution behind the folded BatchNorm multiplies: scale_g * x + shift_g per group of images."""
    xd = x[..., :cin].double().cpu().permute(0, 3, 1, 2)
    per = x.shape[0] // groups
    return torch.cat([xd[i * per:(i + 1) * per] * scale[i, :cin].double().cpu().view(1, -1, 1, 1) + shift[i, :cin].double().cpu().view(1, -1, 1, 1) for i in range(groups)])


@pytest.mark.parametrize("n,cin,cout,H,W,groups,drop", [(48, 256, 256, 32, 32, 12, 0.2), (24, 128, 256, 32, 32, 4, 0.0), (12, 64, 128, 16, 48, 3, 0.5), (16, 128, 256, 64, 64, 2, 0.0),
                                                         (12, 64, 128, 18, 48, 3, 0.0)])
def test_weight_gradient_of_a_pooled_gradient_on_the_sparse_matrix_instruction(device, n, cin, cout, H, W, groups, drop):
    """Round 5: sf_conv3x3_bwd_weight_folded_sparse24 - dout = the gradient behind a 2x2 max-pooling (one non-zero per window and channel, fewer after the
    dropout) as the SPARSE operand of v_smfmac_f32_32x32x32_bf16 (two dout rows per instruction, compressed in registers).  Same products as the dense
    kernel: dW, db and the BatchNorm-backward sums against the dense launch at fp32 summation distance, dW against float64 of exactly the operands."""
    import satflow_amd
    from satflow_amd import kernels as K
    from satflow_amd._hip import T, cpad, lib
    from satflow_amd.functional import ConvEngine

    satflow_amd.set_compute_dtype("bf16a")
    try:
        g = torch.Generator().manual_seed(n + cin)
        eng = ConvEngine([cin], cout)
        gm = eng.wgrad_map
        assert lib().sf_conv3x3_bwd_weight_folded_sparse24_supported(eng.coutp, cpad(cin), n, H, W, groups), "this shape must take the sparse path"
        x = torch.randn(n, H, W, cpad(cin), generator=g).to(device).to(torch.bfloat16)
        x[..., cin:] = 0
        y = torch.randn(n, H, W, eng.coutp, generator=g).to(device).to(torch.bfloat16)
        pooled, route = K.maxpool2_route_fwd(y, None, torch.bfloat16, None)
        gp = torch.randn(pooled.shape, generator=g).to(device).to(torch.bfloat16)
        if drop:
            gp = gp * (torch.rand(gp.shape, generator=g).to(device) >= drop).to(torch.bfloat16)   # zeros inside the structure (dropped positions) are fine
        dout = K.maxpool2_route_bwd(route, gp, tuple(y.shape), torch.bfloat16, None, None)
        nz = (dout.view(n, H, W // 2, 2, -1) != 0).sum(3)
        assert int(nz.max()) <= 1, "at most one non-zero per aligned horizontal pixel pair and channel"
        scale = (0.5 + torch.rand(groups, cpad(cin), generator=g)).to(device)
        shift = torch.randn(groups, cpad(cin), generator=g).to(device)
        w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(device)
        mean, rstd = torch.randn(groups, cpad(cin), generator=g).to(device), (0.5 + torch.rand(groups, cpad(cin), generator=g)).to(device)
        res = []
        for sparse in (False, True):
            dw, db = torch.full_like(w, float("nan")), torch.full((cout,), float("nan"), device=device)
            sums = torch.full((groups, 2, cpad(cin)), float("nan"), dtype=torch.float64, device=device)
            K.conv3x3_bwd_weight_folded(T(x), T(dout), n, H, W, gm, scale, shift, dw, db, bn=(w, mean, rstd, sums), pooled_gradient=sparse)
            res.append((dw, db, sums))
        torch.cuda.synchronize()
        (dw0, db0, s0), (dw1, db1, s1) = res
        assert torch.isfinite(dw1).all() and torch.isfinite(db1).all() and torch.isfinite(s1).all()
        assert rel_l2(dw1, dw0) < 5e-6 and rel_l2(db1, db0) < 1e-6, (rel_l2(dw1, dw0), rel_l2(db1, db0))
        assert float((s1 - s0).norm() / s0.norm()) < 5e-6
        # the POOLED form: the sparse operand built from the pooled gradient + the routing record (what sf_maxpool2_route_bwd was given) instead of from
        # dout - with and without the pooling's outer image permutation, and with the dropout masks applied by the routing kernel (its masked side output)
        if K.conv3x3_bwd_weight_pooled_supported(eng.coutp, cpad(cin), n, H, W, groups):
            for perm, dropout in ((None, None), ((groups, 2), None) if n % (2 * groups) == 0 else (None, None), (None, (0.3, 0.25, gp[: n // groups].numel(), 77, 99))):
                masked = torch.full_like(gp, float("nan")) if dropout is not None else None
                if perm is not None:   # the pooled tensor lives in the pooling's OUTPUT image order
                    L, Tt = perm
                    B = n // (L * Tt)
                    gp_in = gp.view(L, Tt, B, *gp.shape[1:]).transpose(0, 1).reshape(gp.shape).contiguous()
                else:
                    gp_in = gp
                dout_p = K.maxpool2_route_bwd(route, gp_in, tuple(y.shape), torch.bfloat16, perm, dropout, masked)
                if dropout is None:
                    assert torch.equal(dout_p, dout)
                else:
                    assert torch.isfinite(masked.float()).all() and int((masked == 0).sum()) > masked.numel() // 4
                ref_w, ref_b = torch.full_like(w, float("nan")), torch.full((cout,), float("nan"), device=device)
                ref_s = torch.full((groups, 2, cpad(cin)), float("nan"), dtype=torch.float64, device=device)
                K.conv3x3_bwd_weight_folded(T(x), T(dout_p), n, H, W, gm, scale, shift, ref_w, ref_b, bn=(w, mean, rstd, ref_s), pooled_gradient=True)
                dw2, db2 = torch.full_like(w, float("nan")), torch.full((cout,), float("nan"), device=device)
                s2 = torch.full((groups, 2, cpad(cin)), float("nan"), dtype=torch.float64, device=device)
                K.conv3x3_bwd_weight_folded(T(x), T(dout_p), n, H, W, gm, scale, shift, dw2, db2, bn=(w, mean, rstd, s2), pooled_gradient=True,
                                            pooled=(masked if masked is not None else gp_in, route, perm))
                torch.cuda.synchronize()
                assert torch.isfinite(dw2).all() and torch.isfinite(db2).all() and torch.isfinite(s2).all(), (perm, dropout)
                # the same non-zero products (an empty pixel pair carries another position code: its zero enters the instruction's adder elsewhere)
                assert rel_l2(dw2, ref_w) < 1e-6 and float((s2 - ref_s).norm() / ref_s.norm()) < 1e-6, (perm, dropout, rel_l2(dw2, ref_w))
                assert rel_l2(db2, ref_b) < 1e-6, (perm, dropout, rel_l2(db2, ref_b))
                # ... and the float64 leg of THIS form: the oracle's dout is the (masked) pooled gradient scattered through torch's own
                # max_pool2d indices of y (first maximum in row-major order, the kernels' rule) - neither the routing record nor sf_maxpool2_route_bwd's
                # scatter enters it; then conv2d's weight gradient, the bias gradient and the two BatchNorm-backward reductions of exactly those operands
                assert dropout is None or perm is None
                g_in = (masked if masked is not None else gp).double().cpu()                     # pooled gradient in the pooling's INPUT image order
                _, idx = torch.nn.functional.max_pool2d(y.double().cpu().permute(0, 3, 1, 2), 2, return_indices=True)
                d64 = torch.zeros(n, eng.coutp, H * W, dtype=torch.float64).scatter_(2, idx.flatten(2), g_in.permute(0, 3, 1, 2).flatten(2)).view(n, -1, H, W)
                xl = xn64(x, scale, shift, cin, groups).requires_grad_()
                wl = w.double().cpu().requires_grad_()
                gw2, gx2 = torch.autograd.grad(torch.nn.functional.conv2d(xl, wl, None, padding=1), (wl, xl), d64[:, :cout])
                e_w, e_b = rel_l2(dw2.cpu().double(), gw2), rel_l2(db2.cpu().double(), d64[:, :cout].sum(dim=(0, 2, 3)))
                # dn = conv^T(dout, W) enters the BatchNorm = the gradient wrt the convolution's (normalised) input; xhat from the given mean / rstd
                per = n // groups
                dn = gx2.view(groups, per, cin, H, W)
                xh = (x[..., :cin].double().cpu().permute(0, 3, 1, 2).reshape(groups, per, cin, H, W) - mean[:, :cin].double().cpu().view(groups, 1, cin, 1, 1)) \
                    * rstd[:, :cin].double().cpu().view(groups, 1, cin, 1, 1)
                s_ref = torch.stack([dn.sum(dim=(1, 3, 4)), (dn * xh).sum(dim=(1, 3, 4))], 1)
                e_s = float((s2[..., :cin].cpu() - s_ref).norm() / s_ref.norm())
                print(f"pooled sparse wgrad vs float64 (perm {perm}, dropout {dropout is not None}): dW {e_w:.2e} db {e_b:.2e} sums {e_s:.2e}")
                assert e_w < 2e-6 and e_b < 2e-6 and e_s < 2e-6, (perm, dropout, e_w, e_b, e_s)   # observed on MI355X: 0.7 .. 1.1e-7 each
        # float64: dW = sum_g scale_g (.) dWraw_g + shift_g (x) V_g  ==  the weight gradient of conv(scale_g * x + shift_g) for the group's images
        xd = x[..., :cin].double().cpu().permute(0, 3, 1, 2)
        per = n // groups
        xn = torch.cat([xd[i * per:(i + 1) * per] * scale[i, :cin].double().cpu().view(1, -1, 1, 1) + shift[i, :cin].double().cpu().view(1, -1, 1, 1) for i in range(groups)])
        wd = w.double().cpu().requires_grad_()
        out = torch.nn.functional.conv2d(xn, wd, None, padding=1)
        (gw,) = torch.autograd.grad(out, wd, dout[..., :cout].double().cpu().permute(0, 3, 1, 2))
        assert rel_l2(dw1.cpu().double(), gw) < 2e-5, rel_l2(dw1.cpu().double(), gw)
    finally:
        satflow_amd.set_compute_dtype("f32")
