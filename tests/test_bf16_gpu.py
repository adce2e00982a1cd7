"""GPU: the bf16-MFMA compute mode (bf16 operands, fp32 accumulate, fp32 storage).

Two checks per kernel: (1) EXACTNESS of the kernel's own arithmetic - against the fp32 oracle evaluated on
bf16-rounded inputs and weights the result must agree to fp32-accumulation noise (rtol 1e-4); (2) the distance
to the plain fp32 oracle, i.e. the price of bf16 operands, asserted at 2e-2 of the tensor scale and printed.
"""
import pytest
import torch
import torch.nn.functional as TF

import satflow_amd
from conftest import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture()
def bf16_mode():
    satflow_amd.set_compute_dtype("bf16")
    yield
    satflow_amd.set_compute_dtype("f32")


def _r(t):
    return t.bfloat16().float()


SPLITK_SHAPES = [(256, 256, 2, 20, 20), (512, 128, 1, 16, 16), (272, 256, 2, 36, 24), (1024, 512, 2, 8, 8)]   # few small images, many channels: sf_conv3x3_fwd_splitk


@pytest.mark.parametrize("cin,cout,n,h,w", [(16, 32, 2, 16, 16), (12, 5, 1, 7, 9), (64, 160, 2, 40, 33), (256, 256, 2, 32, 32), (48, 96, 1, 64, 20)] + SPLITK_SHAPES)
def test_conv3x3_bf16(device, bf16_mode, cin, cout, n, h, w):
    from satflow_amd.functional import ConvEngine, conv3x3, nchw_to_nhwc, nhwc_to_nchw

    if (cin, cout, n, h, w) in SPLITK_SHAPES:   # make sure these cases DO run the channel-sliced kernel (forward and input gradient)
        from satflow_amd._hip import SF_BF16, cpad, lib
        eng = ConvEngine([cin], cout)
        assert lib().sf_conv3x3_fwd_splitk_workspace_bytes(n, h, w, eng.fwd_map.Np, eng.fwd_map.nf, cpad(cin), SF_BF16) > 0

    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin**0.5)
    b = torch.randn(cout, generator=g)
    cot = torch.randn(n, cout, h, w, generator=g)
    xr, wr = _r(x).requires_grad_(), _r(wt).requires_grad_()
    ref = TF.conv2d(xr, wr, b, padding=1)
    xd, wd, bd = (t.to(device).requires_grad_() for t in (x, wt, b))
    y = nhwc_to_nchw(conv3x3(ConvEngine([cin], cout), nchw_to_nhwc(xd), wd, bd), cout)
    assert_close(y, ref, "bf16 conv vs oracle on bf16-rounded operands")
    # input gradient: bf16(cotangent) x bf16(W^T)
    (y * cot.to(device)).sum().backward()
    dx_ref = torch.autograd.grad(TF.conv2d(xr, wr, None, padding=1), xr, _r(cot))[0]
    assert_close(xd.grad, dx_ref, "bf16 dgrad", grad=True)
    # weight gradient: bf16(cotangent)^T x bf16(x), fp32 accumulate; bias gradient summed in fp32 from the unrounded cotangent
    dw_ref = torch.autograd.grad(TF.conv2d(xr, wr, None, padding=1), wr, _r(cot))[0]
    assert_close(wd.grad, dw_ref, "bf16 wgrad", grad=True)
    assert_close(bd.grad, cot.sum(dim=(0, 2, 3)), "db", grad=True)
    xf, wf = x.clone().requires_grad_(), wt.clone().requires_grad_()
    full = TF.conv2d(xf, wf, b, padding=1)
    full.backward(cot)
    rel = float((wd.grad.cpu() - wf.grad).norm() / wf.grad.norm())
    print(f"   bf16 dW rel L2 vs fp32 {rel:.3e}")
    assert rel < 1e-2
    scale = float(full.abs().max())
    err = float((y.detach().cpu() - full.detach()).abs().max())
    print(f"bf16 conv cin={cin}: max abs err vs fp32 {err:.3e} (scale {scale:.2f})")
    assert err < 2e-2 * scale


@pytest.mark.parametrize("case", ["a", "b", "odd", "hot"])
def test_cell_bf16_vs_golden(device, bf16_mode, case):
    """ConvLSTM cell in bf16 mode: exact vs oracle-on-rounded-operands; close to the reference golden."""
    import os

    import numpy as np
    from conftest import GOLDEN
    from oracle import convlstm as O
    from satflow_amd.models.layers import ConvLSTMCell

    G = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, f"convlstm_cell_{case}.npz")).items()}
    hid, cin = G["h"].shape[1], G["x"].shape[1]
    cell = ConvLSTMCell(cin, hid, (3, 3), True).to(device)
    with torch.no_grad():
        cell.conv.weight.copy_(G["weight"])
        cell.conv.bias.copy_(G["bias"])
    h1, c1 = cell(G["x"].to(device), (G["h"].to(device), G["c"].to(device)))
    rh, rc = O.convlstm_cell(_r(G["x"]), _r(G["h"]), G["c"], _r(G["weight"]), G["bias"])
    assert_close(h1, rh, "h' (rounded-operand oracle)")
    assert_close(c1, rc, "c' (rounded-operand oracle)")
    err = max(float((h1.cpu() - G["h_out"]).abs().max()), float((c1.cpu() - G["c_out"]).abs().max()))
    print(f"bf16 cell {case}: max abs err vs reference golden {err:.3e}")
    assert err < 3e-2


@pytest.mark.parametrize("case", ["cfg1_h8", "cfg1_h32_hot", "rect_h16_o12"])
def test_model_bf16_vs_golden(device, bf16_mode, case):
    """Whole ConvLSTM in bf16 mode vs the reference golden: error of the same order as the reference's own
    bf16-autocast run (stored in the golden as pred_bf16_autocast)."""
    from test_convlstm_gpu import _load, _model_from_golden

    G = _load(f"convlstm_model_{case}.npz")
    m, fs = _model_from_golden(G, device)
    x = G["x"].to(device).requires_grad_()
    pred = m(x, fs)
    (pred * G["cot"].to(device)).sum().backward()
    ours = float((pred.detach().cpu() - G["pred"]).abs().max())
    theirs = float((G["pred_bf16_autocast"] - G["pred"]).abs().max())
    print(f"bf16 model {case}: max abs err ours {ours:.3e} vs reference autocast {theirs:.3e}")
    assert ours <= max(2 * theirs, 5e-3)
    g = dict(m.model.named_parameters())["decoder_CNN.weight"].grad.cpu()
    ref = G["grad.decoder_CNN.weight"]
    rel = float((g - ref).norm() / ref.norm())
    print(f"   decoder_CNN.weight grad rel L2 {rel:.3e}")
    assert rel < 5e-2


def test_metnet_bf16_train_step(device, bf16_mode):
    """Whole MetNet training step in bf16-MFMA mode against the fp32 oracle: output and every parameter gradient within
    bf16-operand tolerance (relative L2), printed for the record."""
    from oracle import metnet as M
    from test_metnet_gpu import _g, _metnet_pair

    cfg = dict(input_channels=13, sat_channels=12, input_size=16, output_channels=3, hidden_dim=32, forecast_steps=4)
    net, P = _metnet_pair(device, cfg)
    x = torch.randn(2, 3, 13, 64, 64, generator=_g(51))
    cot = torch.randn(2, 4, 3, 4, 4, generator=_g(52))
    ref = M.metnet_forward(x, P, sat_channels=12, input_size=16, forecast_steps=4)
    (ref * cot).sum().backward()
    net.train()
    out = net(x.to(device))
    (out * cot.to(device)).sum().backward()
    # yardstick: the same oracle under torch.autocast(bfloat16) on the CPU, i.e. what the reference's own bf16 path costs.
    # (deep encoder gradients are ill-conditioned here: 3 training-mode BatchNorms over ~100 samples per channel and two
    # max-pools whose argmax can flip under bf16 rounding)
    P2 = {k: v.detach().clone().requires_grad_() for k, v in P.items()}
    with torch.autocast("cpu", dtype=torch.bfloat16):
        ref16 = M.metnet_forward(x, P2, sat_channels=12, input_size=16, forecast_steps=4)
    (ref16.float() * cot).sum().backward()
    rel = lambda a, b: float((a.detach().cpu().float() - b.detach()).norm() / (b.detach().norm() + 1e-30))
    ours_out, theirs_out = rel(out, ref), rel(ref16, ref)
    assert ours_out < max(2 * theirs_out, 2e-2), (ours_out, theirs_out)
    worst = ("out", 0.0, 0.0)
    for k, p in net.named_parameters():
        if k in ("image_encoder.module.module.0.bias", "image_encoder.module.module.4.bias", "image_encoder.module.module.6.bias"):
            continue  # convolution biases in front of a training-mode BatchNorm: the true gradient is zero
        ours, theirs = rel(p.grad, P[k].grad), rel(P2[k].grad, P[k].grad)
        if ours > worst[1]:
            worst = (k, ours, theirs)
        assert ours < max(2 * theirs, 3e-2), (k, ours, theirs)
    print(f"bf16 MetNet step: output rel L2 ours {ours_out:.2e} / CPU autocast {theirs_out:.2e}; worst gradient {worst[0]}: ours {worst[1]:.2e} / autocast {worst[2]:.2e}")


def test_conv_broadcast_sources(device):
    """Two-source convolution with the image-index remap of sfTensor (a source shared by all lead times + a per-lead
    constant source), fp32 mode, against conv2d on the materialised concatenation."""
    import satflow_amd
    from satflow_amd.functional import ConvEngine, conv3x3_broadcast, nchw_to_nhwc, nhwc_to_nchw

    satflow_amd.set_compute_dtype("f32")
    g = torch.Generator().manual_seed(7)
    F_, L, c0, c1, co, S = 3, 4, 20, 4, 24, 12
    frames = torch.randn(F_, c0, S, S, generator=g)
    planes = torch.randn(L, c1, S, S, generator=g)
    w = torch.randn(co, c0 + c1, 3, 3, generator=g) * 0.1
    b = torch.randn(co, generator=g)
    full = torch.cat([torch.cat((frames, planes[l : l + 1].expand(F_, -1, -1, -1)), 1) for l in range(L)], 0)
    wr = w.clone().requires_grad_()
    ref = TF.conv2d(full, wr, b, padding=1)
    cot = torch.randn(ref.shape, generator=g)
    (ref * cot).sum().backward()
    wd = w.to(device).requires_grad_()
    y = conv3x3_broadcast(ConvEngine([c0, c1], co), nchw_to_nhwc(frames.to(device)), nchw_to_nhwc(planes.to(device)), wd, b.to(device),
                          L * F_, (0, F_), (F_, 0))
    y = nhwc_to_nchw(y, co)
    (y * cot.to(device)).sum().backward()
    assert_close(y, ref, "broadcast conv")
    assert_close(wd.grad, wr.grad, "broadcast conv dW", grad=True)


@pytest.mark.parametrize("storage", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,cout,n,h,w,groups", [(16, 40, 4, 16, 16, 2), (160, 256, 6, 32, 32, 3), (32, 48, 2, 40, 20, 1)])
def test_conv_epilogue_statistics_feed_batchnorm(device, storage, cin, cout, n, h, w, groups):
    """`sf_conv3x3_fwd_stats` + `sf_batchnorm_train_fwd_stats` == convolution followed by the BatchNorm that computes its own
    statistics (same kernels otherwise): the per-tile sums replace the statistics pass, nothing else changes."""
    import satflow_amd
    from satflow_amd.functional import ConvEngine, batchnorm, conv3x3, nchw_to_nhwc

    satflow_amd.set_compute_dtype("bf16a" if storage == torch.bfloat16 else "bf16")
    try:
        g = torch.Generator().manual_seed(cin + h)
        x = nchw_to_nhwc(torch.randn(n, cin, h, w, generator=g).to(device)).to(storage)
        wt = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin**0.5)).to(device)
        b = torch.randn(cout, generator=g).to(device)
        eng = ConvEngine([cin], cout)
        bn1, bn2 = torch.nn.BatchNorm2d(cout).to(device), torch.nn.BatchNorm2d(cout).to(device)
        y1, st = conv3x3(eng, x, wt, b, out_dtype=storage, want_stats=True)
        assert st is not None and st.data.shape == (n * st.tiles, eng.fwd_map.Np, 2)
        y2 = conv3x3(eng, x, wt, b, out_dtype=storage)
        from satflow_amd._hip import SF_BF16, cpad, lib
        if lib().sf_conv3x3_fwd_splitk_workspace_bytes(n, h, w, eng.fwd_map.Np, eng.fwd_map.nf, cpad(cin), SF_BF16):
            # few images: the plain launch sums its input channels in slices (sf_conv3x3_fwd_splitk) - same products, another fp32 order
            d = (y1.float() - y2.float()).abs()
            assert float((d > 2.0**-7 * y2.float().abs() + 1e-5).float().mean()) == 0.0
            y2 = y1
        else:
            assert torch.equal(y1, y2)
        o1, o2 = batchnorm(y1, bn1, groups, True, st), batchnorm(y2, bn2, groups, True)
        # statistics: fp32 per-tile sums of the same stored values, then fp64 - vs fp32 per-row sums, then fp64
        assert_close(bn1.running_mean, bn2.running_mean, "running_mean from epilogue statistics", rtol=1e-5, atol=1e-6)
        assert_close(bn1.running_var, bn2.running_var, "running_var from epilogue statistics", rtol=1e-5, atol=1e-6)
        if storage == torch.float32:
            assert_close(o1, o2, "BatchNorm output", rtol=1e-5, atol=1e-5)
        else:
            d = (o1.float() - o2.float()).abs()
            assert float((d > 2.0**-7 * o2.float().abs() + 1e-6).float().mean()) == 0.0  # at most 1 bf16 ulp apart
    finally:
        satflow_amd.set_compute_dtype("f32")


@pytest.mark.parametrize("storage", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,cout,n,h,w", [(32, 128, 513, 16, 16), (16, 256, 512, 9, 12)])
def test_conv3x3_dual_image_tiles(device, storage, cin, cout, n, h, w):
    """Many small images with >= 128 output channels take the dual-image layout of the 8-wave kernel (two images per 32x16
    tile; odd image count: the last tile is half empty).  Forward and input gradient against the oracle on rounded operands."""
    import satflow_amd
    from satflow_amd.functional import ConvEngine, conv3x3, nchw_to_nhwc, nhwc_to_nchw

    satflow_amd.set_compute_dtype("bf16a" if storage == torch.bfloat16 else "bf16")
    try:
        g = torch.Generator().manual_seed(n + cin)
        x = torch.randn(n, cin, h, w, generator=g)
        wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin**0.5)
        b = torch.randn(cout, generator=g)
        cot = torch.randn(n, cout, h, w, generator=g)
        xr, wr = _r(x).requires_grad_(), _r(wt)
        ref = TF.conv2d(xr, wr, b, padding=1)
        dx_ref = torch.autograd.grad(TF.conv2d(xr, wr, None, padding=1), xr, _r(cot))[0]
        xd = nchw_to_nhwc(x.to(device)).to(storage).requires_grad_()
        y = conv3x3(ConvEngine([cin], cout), xd, wt.to(device), b.to(device), out_dtype=storage)
        y.backward(nchw_to_nhwc(cot.to(device)).to(storage))  # dgrad: cout -> cin (NF = 1: not dual) - checks the pair
        yn, dxn = nhwc_to_nchw(y.detach().float().contiguous(), cout).cpu(), nhwc_to_nchw(xd.grad.float().contiguous(), cin).cpu()
        if storage == torch.float32:
            assert_close(yn, ref, "dual-tile conv")
            assert_close(dxn, dx_ref, "its dgrad", grad=True)
        else:
            for got, want in ((yn, ref.detach()), (dxn, dx_ref)):
                scale = float(want.abs().max())
                assert float(((got - want).abs() - (2.0**-8 * want.abs() + 1e-6 * scale)).max()) <= 0
    finally:
        satflow_amd.set_compute_dtype("f32")
