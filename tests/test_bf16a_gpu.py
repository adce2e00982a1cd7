"""GPU: "bf16a" mode - bf16 kernels AND bf16 storage of the MetNet image encoder's activations / gradients.

Every kernel that accepts bf16-stored tensors (sfTensor.dtype = SF_BF16) is checked against the SAME computation on
the fp32 widening of the same bf16 inputs: the arithmetic is fp32 (or exact-product bf16 MFMA with fp32 accumulate)
either way, so the only admissible difference is the final round-to-nearest-even to bf16, i.e. <= 1 bf16 ulp
(2^-8 relative) plus accumulation-order noise.  Max pooling must agree exactly.  The whole-model test uses the CPU
autocast oracle as the yardstick, as tests/test_bf16_gpu.py does.
"""
import pytest
import torch
import torch.nn.functional as TF

import satflow_amd
from conftest import assert_close

pytestmark = pytest.mark.gpu

ULP = 2.0**-8


@pytest.fixture()
def bf16a_mode():
    satflow_amd.set_compute_dtype("bf16a")
    yield
    satflow_amd.set_compute_dtype("f32")


def _r(t):
    return t.bfloat16().float()


def _within_ulp(got, ref, what, extra=1e-6):
    """got (bf16 storage) vs ref (fp32): one bf16 ulp of the value + `extra` of the tensor scale (accumulation noise that can
    move a value across a rounding boundary is covered by the full ulp; a half ulp is the rounding itself)."""
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = float(ref.abs().max()) + 1e-30
    excess = (got - ref).abs() - (ULP * ref.abs() + extra * scale)
    assert float(excess.max()) <= 0, f"{what}: exceeds 1 bf16 ulp by {float(excess.max()):.3e} (scale {scale:.3e})"


def _nhwc_b(x, device):
    """NCHW fp32 (CPU) -> padded NHWC bf16 leaf on the device, plus its exact fp32 NCHW widening on the CPU."""
    from satflow_amd.functional import nchw_to_nhwc

    xb = nchw_to_nhwc(x.to(device)).bfloat16().requires_grad_()
    return xb, _r(x)


def _nchw(t, c):
    from satflow_amd.functional import nhwc_to_nchw

    return nhwc_to_nchw(t.detach().float().contiguous(), c).cpu()


@pytest.mark.parametrize("cin,cout,n,h,w", [(16, 32, 2, 16, 16), (12, 5, 1, 7, 9), (64, 160, 2, 40, 33), (256, 256, 2, 32, 32), (160, 256, 3, 24, 40),
                                            (256, 256, 2, 20, 20), (512, 128, 1, 16, 16)])   # the last two: channel-sliced launches (sf_conv3x3_fwd_splitk), bf16 in and out
def test_conv3x3_bf16_storage(device, bf16a_mode, cin, cout, n, h, w):
    from satflow_amd.functional import ConvEngine, conv3x3

    g = torch.Generator().manual_seed(cin * 7 + cout)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin**0.5)
    b = torch.randn(cout, generator=g)
    cot = torch.randn(n, cout, h, w, generator=g)
    xb, xr = _nhwc_b(x, device)
    cb, cr = _nhwc_b(cot, device)
    xr.requires_grad_()
    wr = _r(wt).requires_grad_()
    ref = TF.conv2d(xr, wr, b, padding=1)
    wd, bd = wt.to(device).requires_grad_(), b.to(device).requires_grad_()
    y = conv3x3(ConvEngine([cin], cout), xb, wd, bd, out_dtype=torch.bfloat16)
    assert y.dtype == torch.bfloat16
    _within_ulp(_nchw(y, cout), ref, "bf16-stored conv output")
    y.backward(cb.detach())
    assert xb.grad.dtype == torch.bfloat16
    dx_ref, dw_ref = torch.autograd.grad(TF.conv2d(xr, wr, None, padding=1), (xr, wr), cr)
    _within_ulp(_nchw(xb.grad, cin), dx_ref, "bf16-stored dgrad")
    assert_close(wd.grad, dw_ref, "wgrad from bf16-stored tensors", grad=True)
    assert_close(bd.grad, cr.sum(dim=(0, 2, 3)), "db from bf16-stored cotangent", grad=True)


def test_conv3x3_bf16_source_fp32_out(device, bf16a_mode):
    """bf16-stored source, fp32 output (the storage type is per tensor)."""
    from satflow_amd.functional import ConvEngine, conv3x3

    g = torch.Generator().manual_seed(5)
    x, wt = torch.randn(2, 48, 20, 18, generator=g), torch.randn(40, 48, 3, 3, generator=g) * 0.05
    xb, xr = _nhwc_b(x, device)
    y = conv3x3(ConvEngine([48], 40), xb, wt.to(device), None)
    assert y.dtype == torch.float32
    assert_close(_nchw(y, 40), TF.conv2d(xr, _r(wt), None, padding=1), "bf16 source -> fp32 output")


def test_conv3x3_bf16_kernel_rejects_misaligned_output(device, bf16a_mode):
    """The SF_BF16 kernel writes 16-byte channel groups: a channel-slice output that starts on a 4-channel boundary of a
    bf16 tensor (8 bytes) must be refused by the C entry, not written with misaligned stores."""
    from satflow_amd import kernels as K
    from satflow_amd._hip import NULL, T
    from satflow_amd.functional import ConvEngine

    eng = ConvEngine([16], 16)
    w = torch.randn(16, 16, 3, 3, device=device); b = torch.randn(16, device=device)
    packed, bp = K.pack_weights(w, b, eng.fwd_map, False)
    x = torch.randn(1, 8, 8, 16, device=device).bfloat16()
    y = torch.zeros(1, 8, 8, 32, device=device, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="16-byte aligned output"):
        K.conv3x3(T(x), NULL, 1, 8, 8, packed, bp, eng.fwd_map, T(y, 16, 4))
    K.conv3x3(T(x), NULL, 1, 8, 8, packed, bp, eng.fwd_map, T(y, 16, 8))  # an 8-channel (16-byte) offset is fine
    torch.cuda.synchronize()
    assert float(y[..., 8:24].float().abs().max()) > 0 and float(y[..., :8].float().abs().max()) == 0


def test_bf16_storage_needs_bf16_kernels(device):
    from satflow_amd.functional import ConvEngine, conv3x3

    satflow_amd.set_compute_dtype("f32")
    xb, _ = _nhwc_b(torch.randn(1, 16, 8, 8), device)
    with pytest.raises(RuntimeError, match="bf16"):
        conv3x3(ConvEngine([16], 16), xb, torch.randn(16, 16, 3, 3, device=device), None)
    # entry points without a bf16 instantiation refuse bf16 storage instead of misreading it
    from satflow_amd import kernels as K

    with pytest.raises(RuntimeError, match="bf16"):
        K.linear_fwd(xb.detach(), torch.randn(16, 16, device=device), None, 16)


@pytest.mark.parametrize("perm", [None, (3, 2)])
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_maxpool_bf16_storage(device, perm, out_dtype):
    from satflow_amd.functional import maxpool2

    g = torch.Generator().manual_seed(11)
    xb = torch.randn(12, 10, 14, 32, generator=g).to(device).bfloat16().requires_grad_()
    xf = xb.detach().float().requires_grad_()
    yb, yf = maxpool2(xb, perm, out_dtype=out_dtype), maxpool2(xf, perm)
    assert yb.dtype == out_dtype
    assert torch.equal(yb.float(), yf), "max pooling of bf16 values is exact"
    cot = torch.randn(yf.shape, generator=g).to(device).to(out_dtype)
    yb.backward(cot)
    yf.backward(cot.float())
    assert xb.grad.dtype == torch.bfloat16
    assert torch.equal(xb.grad, xf.grad.bfloat16()), "routing is exact; an fp32 cotangent is rounded once on the way into the bf16 gradient"


def test_batchnorm_bf16_storage(device):
    from satflow_amd.functional import batchnorm

    g = torch.Generator().manual_seed(12)
    C, groups = 48, 3
    xb = (torch.randn(6, 9, 7, C, generator=g) * 2 + 0.5).to(device).bfloat16().requires_grad_()
    xf = xb.detach().float().requires_grad_()
    bn_b, bn_f = torch.nn.BatchNorm2d(40).to(device), torch.nn.BatchNorm2d(40).to(device)
    with torch.no_grad():
        bn_b.weight.copy_(torch.randn(40, generator=g)); bn_b.bias.copy_(torch.randn(40, generator=g))
    bn_f.load_state_dict(bn_b.state_dict())
    yb, yf = batchnorm(xb, bn_b, groups, True), batchnorm(xf, bn_f, groups, True)
    assert yb.dtype == torch.bfloat16
    _within_ulp(yb[..., :40], yf[..., :40], "BatchNorm output")
    assert_close(bn_b.running_var, bn_f.running_var, "running_var")
    cot = torch.randn(yf.shape, generator=g).to(device).bfloat16()
    yb.backward(cot)
    yf.backward(cot.float())
    _within_ulp(xb.grad[..., :40], xf.grad[..., :40], "BatchNorm dx", extra=1e-5)
    assert_close(bn_b.weight.grad, bn_f.weight.grad, "dgamma", grad=True)
    assert_close(bn_b.bias.grad, bn_f.bias.grad, "dbeta", grad=True)


def test_leadtime_pool_bf16_storage(device):
    from satflow_amd.functional import leadtime_pool

    g = torch.Generator().manual_seed(13)
    Fr, S, C, cimg, L = 4, 12, 32, 20, 5
    base_b = torch.randn(Fr, S, S, C, generator=g).to(device).bfloat16().requires_grad_()
    base_f = base_b.detach().float().requires_grad_()
    w_b = (torch.randn(28, cimg + L, 3, 3, generator=g) * 0.3).to(device).requires_grad_()
    w_f = w_b.detach().clone().requires_grad_()
    pb, pf = leadtime_pool(base_b, w_b, cimg, L), leadtime_pool(base_f, w_f, cimg, L)
    assert pb.dtype == torch.bfloat16
    _within_ulp(pb, pf, "lead-time pooling output", extra=0.0)
    cot = torch.randn(pf.shape, generator=g).to(device).bfloat16()
    pb.backward(cot)
    pf.backward(cot.float())
    _within_ulp(base_b.grad, base_f.grad, "lead-time pooling d(base)", extra=1e-6)
    assert_close(w_b.grad, w_f.grad, "lead-time pooling dW1", grad=True)


def test_preprocess_bf16_storage(device):
    from satflow_amd import kernels as K

    x = torch.randn(2, 3, 13, 32, 32, generator=torch.Generator().manual_seed(14)).to(device)
    fb, ff = K.metnet_preprocess(x, 12, 8, torch.bfloat16), K.metnet_preprocess(x, 12, 8)
    assert fb.dtype == torch.bfloat16 and torch.equal(fb, ff.bfloat16())


def test_metnet_bf16a_train_step(device, bf16a_mode):
    """Whole MetNet training step with bf16-stored encoder activations against the fp32 oracle; yardstick = the same
    oracle under torch.autocast(bfloat16) on the CPU (which also leaves bf16 tensors between the encoder's layers)."""
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
    assert out.dtype == torch.float32
    (out * cot.to(device)).sum().backward()
    P2 = {k: v.detach().clone().requires_grad_() for k, v in P.items()}
    with torch.autocast("cpu", dtype=torch.bfloat16):
        ref16 = M.metnet_forward(x, P2, sat_channels=12, input_size=16, forecast_steps=4)
    (ref16.float() * cot).sum().backward()
    rel = lambda a, b: float((a.detach().cpu().float() - b.detach()).norm() / (b.detach().norm() + 1e-30))
    ours_out, theirs_out = rel(out, ref), rel(ref16, ref)
    assert ours_out < max(2 * theirs_out, 3e-2), (ours_out, theirs_out)
    worst = ("out", 0.0, 0.0)
    for k, p in net.named_parameters():
        if k in ("image_encoder.module.module.0.bias", "image_encoder.module.module.4.bias", "image_encoder.module.module.6.bias"):
            continue  # convolution biases in front of a training-mode BatchNorm: the true gradient is zero
        assert p.grad.dtype == torch.float32
        ours, theirs = rel(p.grad, P[k].grad), rel(P2[k].grad, P[k].grad)
        if ours > worst[1]:
            worst = (k, ours, theirs)
        assert ours < max(2 * theirs, 5e-2), (k, ours, theirs)
    print(f"bf16a MetNet step: output rel L2 ours {ours_out:.2e} / CPU autocast {theirs_out:.2e}; worst gradient {worst[0]}: ours {worst[1]:.2e} / autocast {worst[2]:.2e}")


@pytest.mark.parametrize("case", ["cfg1_h8", "cfg1_h32_hot", "rect_h16_o12", "cfg1_h64_hot"])
def test_convlstm_bf16a_vs_bf16(device, case):
    """ConvLSTM stack with bf16-STORED hidden states, gates and gate gradients ("bf16a") against the same stack with fp32-stored
    ones ("bf16"): the forward pass never reads the stored gates and reads a hidden state only as a bf16 MFMA operand (the
    stored value IS that operand), so predictions are bit-identical; every parameter gradient and the
    input gradient stay within a few bf16 ulps (relative L2) of the fp32-storage run and as close to the reference golden."""
    from test_convlstm_gpu import _load, _model_from_golden

    G = _load(f"convlstm_model_{case}.npz")
    res = {}
    for mode in ("bf16", "bf16a"):
        satflow_amd.set_compute_dtype(mode)
        try:
            m, fs = _model_from_golden(G, device)
            x = G["x"].to(device).requires_grad_()
            pred = m(x, fs)
            (pred * G["cot"].to(device)).sum().backward()
            res[mode] = (pred.detach().cpu(), x.grad.cpu(), {k: p.grad.cpu() for k, p in m.model.named_parameters()})
        finally:
            satflow_amd.set_compute_dtype("f32")
    assert torch.equal(res["bf16"][0], res["bf16a"][0]), "forward must not depend on the gate storage type"
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
    worst = max([rel(res["bf16a"][1], res["bf16"][1])] + [rel(res["bf16a"][2][k], res["bf16"][2][k]) for k in res["bf16"][2]])
    print(f"bf16a ConvLSTM {case}: worst gradient rel L2 vs fp32-stored gates {worst:.2e}")
    assert worst < 1e-2
    for k, g in res["bf16a"][2].items():
        if f"grad.{k}" not in G:  # the large fixture keeps only the small gradients
            continue
        ref = G[f"grad.{k}"]
        assert rel(g, ref) < max(2 * rel(res["bf16"][2][k], ref), 2e-2), k


def test_wgrad_bf16_dout_fp32_sources(device, bf16a_mode):
    """Weight gradient with a bf16-stored output gradient against fp32-stored sources (two sources, the first 16 lanes
    wide: the ConvLSTM cell's layout) == the all-fp32-storage kernel fed the widened gradient."""
    from satflow_amd import kernels as K
    from satflow_amd._hip import T
    from satflow_amd.functional import nchw_to_nhwc

    g = torch.Generator().manual_seed(21)
    n, h, w, c0, c1, co = 3, 20, 24, 12, 64, 256
    x0 = nchw_to_nhwc(torch.randn(n, c0, h, w, generator=g).to(device))
    x1 = nchw_to_nhwc(torch.randn(n, c1, h, w, generator=g).to(device))
    dz_b = nchw_to_nhwc(torch.randn(n, co, h, w, generator=g).to(device)).bfloat16()
    gm = K.lstm_wgrad_map(c0, c1)
    out = []
    for dz in (dz_b, dz_b.float()):
        dw, db = torch.empty(co, c0 + c1, 3, 3, device=device), torch.empty(co, device=device)
        K.conv3x3_bwd_weight(T(x0), T(x1), T(dz), n, h, w, gm, dw, db, False)
        out.append((dw, db))
    assert_close(out[0][0], out[1][0], "dW: bf16 vs widened fp32 dout", grad=True)
    assert_close(out[0][1], out[1][1], "db", grad=True)


@pytest.mark.parametrize("cin,hid,n,h,w", [(12, 64, 2, 32, 32), (4, 8, 2, 20, 24), (64, 64, 1, 40, 33)])
def test_convlstm_cell_bf16_states(device, bf16a_mode, cin, hid, n, h, w):
    """sf_convlstm_cell_fwd with bf16-STORED x / h_prev / h_out and bf16 gates against the same call on the fp32 widening of
    the same (bf16-representable) inputs: the cell state must agree bit for bit (the MFMA operands are identical), the stored
    hidden state and gates must be the round-to-nearest-even of the fp32-stored ones."""
    from satflow_amd._hip import T
    from satflow_amd.models.layers.ConvLSTM import ConvLSTMCell

    g = torch.Generator().manual_seed(5)
    cell = ConvLSTMCell(cin, hid, (3, 3), True)
    with torch.no_grad():
        cell.conv.weight.mul_(4.0)
        cell.conv.bias.uniform_(-1, 1, generator=g)
    cell = cell.to(device)
    eng = cell.engine
    rnd = lambda *s: _r(torch.randn(*s, generator=g)).to(device)
    x, h0 = rnd(n, h, w, eng.cinp), rnd(n, h, w, eng.hidp)
    x[..., cin:] = 0; h0[..., hid:] = 0
    c0 = torch.randn(n, h, w, eng.hidp, generator=g).to(device)
    out = {}
    for st in (torch.float32, torch.bfloat16):
        h1 = torch.empty(n, h, w, eng.hidp, device=device, dtype=st)
        c1 = torch.empty(n, h, w, eng.hidp, device=device)
        gates = torch.empty(n, h, w, 4 * eng.hidp, device=device, dtype=st)
        xs, hs = x.to(st), h0.to(st)  # keep the converted copies alive: a descriptor does not own its tensor
        eng.step(T(xs), hs, c0, n, h, w, h1, c1, gates)
        out[st] = (h1, c1, gates)
    f, b = out[torch.float32], out[torch.bfloat16]
    assert torch.equal(f[1], b[1]), "cell state depends on the storage type of x / h"
    assert torch.equal(f[0].bfloat16()[..., :hid], b[0][..., :hid]), "bf16-stored hidden state is not the rounded fp32 one"
    assert torch.equal(f[2].bfloat16(), b[2]), "bf16-stored gates are not the rounded fp32 ones"
    assert float(f[1].abs().max()) > 0.5 and torch.isfinite(f[0]).all()


@pytest.mark.parametrize("B,T,cin,hid,h,w,layers", [(2, 3, 16, 16, 4, 4, 1), (1, 4, 40, 24, 5, 7, 2), (3, 5, 256, 64, 16, 16, 1)])
def test_convgru_bf16a_vs_bf16(device, B, T, cin, hid, h, w, layers):
    """ConvGRU sequence with bf16-STORED x-part pre-activations, saved gates and gate gradients ("bf16a") against fp32-stored ones
    ("bf16"): states within the bf16 rounding of the x-part; the gradients differ by the rounding of the saved gates (the
    rounding of dgx / dgh themselves is what their consumers' MFMA operands do anyway)."""
    from satflow_amd import functional as F
    from satflow_amd.models.metnet import ConvGRU

    g = lambda s: torch.Generator().manual_seed(s)
    torch.manual_seed(B + cin)
    rnn = ConvGRU(cin, hid, (3, 3), layers)
    with torch.no_grad():
        for name, p in rnn.named_parameters():
            if name.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=g(15)) * 0.3)
    rnn.eval()
    rnn = rnn.to(device)
    x = torch.randn(B, T, cin, h, w, generator=g(16)).to(device)
    cot_last = torch.randn(B, hid, h, w, generator=g(17)).to(device)
    cot_seq = (torch.randn(B, T, hid, h, w, generator=g(18)) * 0.5).to(device)
    res = {}
    try:
        for mode in ("bf16", "bf16a"):
            satflow_amd.set_compute_dtype(mode)
            rnn.zero_grad()
            xd = x.clone().requires_grad_()
            xs = F._ToNHWC.apply(xd, B, T, cin, h, w, (T * cin * h * w, cin * h * w, h * w))
            seq, last = rnn.run(xs, T, B)
            seq_nchw = F._FromNHWC.apply(seq, (B, T, hid, h, w), B, T, hid, h, w, (T * hid * h * w, hid * h * w, h * w))
            last_nchw = F.nhwc_to_nchw(last[-1], hid)
            ((last_nchw * cot_last).sum() + (seq_nchw * cot_seq).sum()).backward()
            res[mode] = (seq_nchw.detach().clone(), {"dx": xd.grad.clone(), **{k: p.grad.clone() for k, p in rnn.named_parameters()}})
    finally:
        satflow_amd.set_compute_dtype("f32")
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
    # the forward pass never reads the saved gates; with the persistent sequence kernel "bf16a" also stores the x-part of the
    # pre-activations as bf16 (what autocast leaves behind conv_zr / conv_h1), so the states agree to bf16 rounding of those
    fwd = rel(res["bf16a"][0], res["bf16"][0])
    assert fwd < 5e-3, fwd
    worst = max(rel(res["bf16a"][1][k], res["bf16"][1][k]) for k in res["bf16"][1])
    print(f"bf16a ConvGRU: states rel L2 vs fp32-stored x-part {fwd:.2e}; worst gradient rel L2 vs fp32-stored gates {worst:.2e}")
    assert worst < 1e-2


@pytest.mark.parametrize("cin,cout,n,h,w", [
    (160, 256, 40, 32, 32),   # last input-channel tile half empty: 256 x 32 slabs (conv2 of the DownSampler)
    (96, 256, 24, 24, 40),    # the same with two input tiles
    (256, 192, 40, 16, 16),   # last output-channel tile half empty: 64 x 128 slabs (the ConvGRU's gates)
    (108, 160, 24, 32, 32),   # ... with a single pair of input tiles and 32 live output channels (conv1)
    (192, 64, 24, 20, 20),    # ... one output tile, three input tiles: one pair + an odd tile on the regular geometry
])
def test_weight_gradient_edge_slabs(device, bf16a_mode, cin, cout, n, h, w):
    """sf_conv3x3_bwd_weight on bf16-stored tensors for channel counts whose last 64-channel input tile / last 128-channel output tile is half
    empty (the wide and the tall slab geometry of conv3x3_wgrad_bf16_dma.hip) against a float64 evaluation of exactly its operands:
    products of bf16 values are exact in fp32, only the summation order differs (2e-5 of the gradient's norm)."""
    from satflow_amd import kernels as K
    from satflow_amd._hip import T, cpad
    from satflow_amd.functional import ConvEngine

    g = torch.Generator().manual_seed(cin * 7 + cout)
    eng = ConvEngine([cin], cout)
    x = torch.zeros(n, h, w, cpad(cin), device=device, dtype=torch.bfloat16)
    x[..., :cin] = torch.randn(n, h, w, cin, generator=g).to(device).to(torch.bfloat16)
    gy = torch.zeros(n, h, w, eng.coutp, device=device, dtype=torch.bfloat16)
    gy[..., :cout] = torch.randn(n, h, w, cout, generator=g).to(device).to(torch.bfloat16)
    dw = torch.full((cout, cin, 3, 3), float("nan"), device=device)
    db = torch.full((cout,), float("nan"), device=device)
    K.conv3x3_bwd_weight(T(x), T(None), T(gy), n, h, w, eng.wgrad_map, dw, db, False)
    wz = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, device=device, requires_grad=True)
    out = torch.nn.functional.conv2d(x[..., :cin].double().permute(0, 3, 1, 2), wz, None, padding=1)
    (ref,) = torch.autograd.grad((out * gy[..., :cout].double().permute(0, 3, 1, 2)).sum(), wz)
    rel = lambda a, b: float((a.double() - b).norm() / b.norm())
    assert torch.isfinite(dw).all() and torch.isfinite(db).all()
    assert rel(dw, ref) < 2e-5, rel(dw, ref)
    assert rel(db, gy[..., :cout].double().sum((0, 1, 2))) < 2e-5
