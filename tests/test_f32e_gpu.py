"""GPU: the "f32e" compute mode on its own (round 6) - fp32-equivalent 3x3 convolutions from three fp16 MFMA products (include/satflow_hip.h SF_F32E,
conv3x3_f32e.hip / conv3x3_wgrad_f32e.hip).  The fp32-gated tests of the other modules all run in this mode too (conftest.FP32_GATED); here: the kernels
against float64 of the UNROUNDED operands across operand magnitudes (what the split + the per-tensor gradient scale are for), the scale word itself, and
the loud failure past fp16's range."""
import pytest
import torch
import torch.nn.functional as TF

from conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _f32e():
    import satflow_amd

    satflow_amd.set_compute_dtype("f32e")
    yield
    satflow_amd.set_compute_dtype("f32")


def _conv_case(device, cin, cout, n, h, w, xs=1.0, ws=1.0, gs=1.0, seed=0):
    """conv3x3 forward + backward through the product path; float64 reference of the unrounded fp32 operands.  Returns relative L2 errors."""
    from satflow_amd.functional import ConvEngine, conv3x3, nchw_to_nhwc, nhwc_to_nchw

    g = torch.Generator().manual_seed(seed + cin * 131 + cout)
    x = torch.randn(n, cin, h, w, generator=g) * xs
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (ws / (3 * cin ** 0.5))
    b = torch.randn(cout, generator=g) * xs * ws
    cot = torch.randn(n, cout, h, w, generator=g) * gs
    xr, wr, br = (t.double().requires_grad_() for t in (x, wt, b))
    ref = TF.conv2d(xr, wr, br, padding=1)
    (ref * cot.double()).sum().backward()
    xd, wd, bd = (t.to(device).requires_grad_() for t in (x, wt, b))
    y = nhwc_to_nchw(conv3x3(ConvEngine([cin], cout), nchw_to_nhwc(xd), wd, bd), cout)
    (y * cot.to(device)).sum().backward()
    return {"y": rel_l2(y.double().cpu(), ref), "dx": rel_l2(xd.grad.double().cpu(), xr.grad), "dW": rel_l2(wd.grad.double().cpu(), wr.grad),
            "db": rel_l2(bd.grad.double().cpu(), br.grad)}, y


@pytest.mark.parametrize("cin,cout,n,h,w", [(16, 32, 2, 16, 16), (12, 5, 1, 7, 9), (64, 160, 2, 20, 33), (256, 256, 4, 32, 32), (128, 96, 3, 16, 16)])
def test_conv_against_float64(device, cin, cout, n, h, w):
    """Forward, input gradient, weight gradient, bias gradient at O(1) operands: every one within 2e-6 relative L2 of float64 (the exact-fp32 kernels
    are at 1e-7 .. 5e-7 here; three fp16 products leave ~2^-22 per product)."""
    err, _ = _conv_case(device, cin, cout, n, h, w)
    assert max(err.values()) < 2e-6, err


@pytest.mark.parametrize("gs", [1e-12, 1e-7, 1e-3, 1.0, 1e4, 1e9])
def test_gradient_operands_are_scaled_into_range(device, gs):
    """The output gradient of a real training step is ~1e-7 (an MSE mean over 10^7 elements) - below fp16's normal range - and a random cotangent of a test
    is O(1) or larger.  The kernels scale a gradient operand by the power of two that puts its amax word (sf_amax / sfTensor.amax) at 2^14 and undo it
    on the accumulators: the relative error of dx and dW does not depend on the gradient's magnitude over 21 orders of magnitude."""
    err, _ = _conv_case(device, 64, 96, 2, 24, 24, gs=gs, seed=3)
    assert err["dx"] < 2e-6 and err["dW"] < 2e-6 and err["db"] < 2e-6, (gs, err)


@pytest.mark.parametrize("xs,ws", [(1e-3, 1.0), (1.0, 1e-2), (30.0, 4.0), (1e3, 1.0)])
def test_forward_operand_magnitudes(device, xs, ws):
    """Forward activations and weights go in unscaled (hi = fp16(x) is normal for |x| in 6e-5 .. 65504, the scaled low part recovers 11 more bits below that
    gracefully): O(1e-3) .. O(1e3) activations, saturating-gate-sized weights."""
    err, _ = _conv_case(device, 48, 64, 2, 16, 16, xs=xs, ws=ws, seed=5)
    assert err["y"] < 2e-6 and err["dx"] < 2e-6 and err["dW"] < 2e-6, (xs, ws, err)


def test_out_of_range_activation_is_loud(device):
    """|x| >= 65520 overflows fp16: the mode does not clamp silently - the affected outputs are non-finite (NaN / inf), which the training loops' loss
    checks and `check_device_errors` users see; the exact-fp32 mode is the tool for such data."""
    from satflow_amd.functional import ConvEngine, conv3x3

    x = torch.randn(1, 8, 8, 16, device=device)
    x[0, 3, 4, 2] = 1.0e5
    w = torch.randn(16, 16, 3, 3, device=device) * 0.1
    y = conv3x3(ConvEngine([16], 16), x, w, None)
    assert not torch.isfinite(y[0, 2:5, 3:6]).all() and torch.isfinite(y[0, 6:, :]).all()


@pytest.mark.parametrize("shape,stride", [((3, 5, 7, 16), 16), ((2, 9, 4, 32), 48), ((1, 1, 1, 4), 4)])
def test_amax_word(device, shape, stride):
    """sf_amax: max |t| of an NHWC channel slice (stride > channels: the lanes beyond are not read), the accumulating second word, zero for zeros."""
    from satflow_amd._hip import T, check, lib, stream_ptr

    torch.manual_seed(shape[-1] + stride)
    full = torch.randn(*shape[:-1], stride, device=device) * 7.0
    full[..., shape[-1]:] = 1e6   # outside the slice
    t = T(full, c=shape[-1])
    word, acc = torch.full((1,), -1.0, device=device), torch.full((1,), 123.0, device=device)
    check(lib().sf_amax(t, full.numel() // stride, word.data_ptr(), acc.data_ptr(), 1, stream_ptr()), "sf_amax")
    want = float(full[..., : shape[-1]].abs().max())
    assert float(word) == want and float(acc) == want
    check(lib().sf_amax(T(torch.zeros_like(full), c=shape[-1]), full.numel() // stride, word.data_ptr(), acc.data_ptr(), 0, stream_ptr()), "sf_amax")
    assert float(word) == 0.0 and float(acc) == want   # the accumulator keeps the larger value when it is not reset


def test_lstm_gate_backward_raises_the_scale_word(device):
    """sf_convlstm_cell_bwd_gates with dz.amax: the word (zeroed by the caller) ends at max |dz| - what sf_amax would have computed in a second pass."""
    from satflow_amd import kernels as K
    from satflow_amd._hip import NULL, T

    n, hid = 2 * 12 * 12, 32
    g = torch.Generator().manual_seed(11)
    mk = lambda c: torch.randn(n, c, generator=g).to(device)
    dh, dc, gates, cp, cn = mk(hid), mk(hid), torch.sigmoid(mk(4 * hid)), mk(hid), mk(hid)
    dz, dcp = torch.empty(n, 4 * hid, device=device), torch.empty(n, hid, device=device)
    word = torch.zeros(1, device=device)
    K.convlstm_cell_bwd_gates([T(dh)], T(dc), T(gates), T(cp), T(cn), n, hid, T(dz, amax=word), T(dcp))
    assert float(word) == float(dz.abs().max()) > 0


def test_f32e_tracks_f32_on_a_recurrent_stack(device):
    """Twelve recurrent steps (errors compound through the gates): predictions and every parameter gradient of the ConvLSTM encoder-decoder in "f32e" within
    1e-5 relative L2 of the exact-fp32 mode on the same weights (hot weights: saturating gates)."""
    import satflow_amd
    from satflow_amd.models import EncoderDecoderConvLSTM

    torch.manual_seed(3)
    m = EncoderDecoderConvLSTM(hidden_dim=32, input_channels=6, out_channels=3, forecast_steps=4).to(device)
    with torch.no_grad():
        for k, p in m.named_parameters():
            p.mul_(4.0) if p.dim() > 1 else p.uniform_(-1, 1)
    x = torch.randn(2, 8, 6, 32, 32, device=device)
    cot = torch.randn(2, 3, 4, 32, 32, device=device)
    res = {}
    for mode in ("f32", "f32e"):
        satflow_amd.set_compute_dtype(mode)
        m.zero_grad()
        y = m(x, 4)
        (y * cot).sum().backward()
        res[mode] = (y.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters()})
    assert rel_l2(res["f32e"][0], res["f32"][0]) < 1e-5
    for k, gref in res["f32"][1].items():
        assert rel_l2(res["f32e"][1][k], gref) < 1e-5, k


def test_gru_gate_backward_raises_both_scale_words(device):
    """sf_convgru_bwd_gates with dgx.amax / dgh.amax: the words end at max |dgx| / max |dgh|; a word shared by two launches keeps the larger value."""
    from satflow_amd import kernels as K
    from satflow_amd._hip import T

    n, hid = 3 * 8 * 8, 16
    g = torch.Generator().manual_seed(17)
    mk = lambda c: torch.randn(n, c, generator=g).to(device)
    gates, hp = torch.sigmoid(mk(4 * hid)), mk(hid)
    words = torch.zeros(2, device=device)
    seen = 0.0
    for scale in (1.0, 0.01):
        dh = mk(hid) * scale
        dgx, dgh = torch.empty(n, 3 * hid, device=device), torch.empty(n, 3 * hid, device=device)
        words[1].zero_()
        K.convgru_bwd_gates([T(dh)], gates, hp, hid, dgx, dgh, None, words[0:1], words[1:2])
        seen = max(seen, float(dgx.abs().max()))
        assert float(words[1]) == float(dgh.abs().max()) and float(words[0]) == seen


@pytest.mark.parametrize("workload", ["metnet", "convlstm"])
def test_bench_workload_steps_stay_finite(device, workload):
    """The benchmark's own training steps in "f32e" (MSE-mean loss: output gradients ~1e-7, dropout, Adam): three steps, finite loss, finite parameters.
    Round 6's first sf_amax read one channel in four; the tests' O(1) cotangents never noticed, the benchmark's MetNet step went NaN - and ran 17 % faster
    on NaN operands, which is how it was found."""
    import bench

    wl = bench.MetNetWorkload(device, 2, 0) if workload == "metnet" else bench.ConvLSTMWorkload(device, 2, 0)
    for _ in range(3):
        loss = wl.step()
    assert torch.isfinite(loss).all(), float(loss)
    assert all(torch.isfinite(p).all() for p in wl.model.parameters())


@pytest.mark.parametrize("xs,bound", [(1e-4, 1e-6), (1e-5, 2e-6), (1e-6, 2e-5), (1e-7, 2e-4)])
def test_tiny_forward_operands_degrade_gracefully(device, xs, bound):
    """Forward operands are NOT scaled (only gradients carry an amax word): below fp16's normal range (6e-5) the high part is a subnormal and the scaled low part
    recovers eleven more bits of what is left - an absolute floor of ~3e-11 per element, i.e. a relative error that grows as the WHOLE tensor shrinks
    (measured on MI355X: |x| ~ 1e-4 / 1e-5 / 1e-6 / 1e-7 -> 1.1e-7 / 6.4e-7 / 6.3e-6 / 6.2e-5 relative L2; the bounds are 3x that).  Tensors of such magnitudes belong in the exact-fp32 mode, or are
    handed over with their own scale word (`T(x, amax=word)`: the kernels honour it on src0 of any launch) - the last assertion."""
    from satflow_amd import kernels as K
    from satflow_amd._hip import NULL, T
    from satflow_amd.functional import ConvEngine

    err, _ = _conv_case(device, 32, 32, 1, 16, 16, xs=xs, seed=9)
    print(f"f32e forward at |x| ~ {xs:g}: rel L2 {err['y']:.2e}")
    assert err["y"] < bound, (xs, err)
    # the same input WITH a scale word: back at the split's own error
    g = torch.Generator().manual_seed(99)
    x = (torch.randn(1, 16, 16, 32, generator=g) * xs).to(device)
    w = (torch.randn(32, 32, 3, 3, generator=g) * 0.1).to(device)
    eng = ConvEngine([32], 32)
    packed, _ = K.pack_weights(w, None, eng.fwd_map, False)
    y = torch.empty(1, 16, 16, 32, device=device)
    K.conv3x3(K.grad_operand(x), NULL, 1, 16, 16, packed, None, eng.fwd_map, T(y))
    ref = TF.conv2d(x.double().cpu().permute(0, 3, 1, 2), w.double().cpu(), padding=1).permute(0, 2, 3, 1)
    assert rel_l2(y.double().cpu(), ref) < 2e-6


def test_batchnorm_backward_raises_the_scale_word_and_the_convolution_takes_it(device, monkeypatch):
    """conv3x3 -> BatchNorm (training) in "f32e": the BatchNorm backward's apply pass raises the scale word of the gradient it writes (``dx.amax``) and tags the
    tensor; the convolution's backward takes the tag instead of running ``sf_amax`` over the tensor again.  Checked: (a) the word equals max |dx| of the
    gradient that reached the convolution, (b) NO sf_amax launch for it, (c) gradients bit-identical to the run with the tag removed (same word either way)."""
    from satflow_amd import kernels as K
    from satflow_amd._hip import lib
    from satflow_amd.functional import ConvEngine, batchnorm, conv3x3

    g = torch.Generator().manual_seed(5)
    n, h, w, cin, cout = 4, 16, 16, 32, 64
    x0 = torch.randn(n, h, w, cin, generator=g).to(device)
    wt0 = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).to(device)
    cot = (torch.randn(n, h, w, cout, generator=g) * 1e-6).to(device)   # a training step's gradient magnitude: unscaled it is below fp16's range
    eng = ConvEngine([cin], cout)

    def run(tagged: bool):
        bn = torch.nn.BatchNorm2d(cout).to(device)
        x, wt = x0.clone().requires_grad_(), wt0.clone().requires_grad_()
        seen, calls = [], []
        real_tag, real_amax = K.tag_amax, lib().sf_amax
        monkeypatch.setattr(K, "tag_amax", lambda t, word: (seen.append((t, word)), real_tag(t, word) if tagged else t)[1])

        class Lib:   # counts sf_amax launches, everything else passes through
            def __getattr__(self, name):
                f = getattr(lib(), name)
                if name != "sf_amax":
                    return f
                return lambda *a: (calls.append(1), f(*a))[1]
        monkeypatch.setattr(K, "lib", lambda: Lib())
        y = batchnorm(conv3x3(eng, x, wt, None), bn, 2, True)
        (y[..., :cout] * cot).sum().backward()
        torch.cuda.synchronize()
        monkeypatch.setattr(K, "lib", lib)
        monkeypatch.setattr(K, "tag_amax", real_tag)
        assert real_amax is lib().sf_amax
        return x.grad, wt.grad, seen, len(calls)

    gx1, gw1, seen, calls1 = run(True)
    assert len(seen) == 1 and seen[0][1] is not None
    dx, word = seen[0]
    assert float(word) == float(dx.abs().max()) > 0
    assert calls1 == 0, calls1
    gx0, gw0, _, calls0 = run(False)
    assert calls0 == 1, calls0
    assert torch.equal(gx1, gx0) and torch.equal(gw1, gw0)
    assert torch.isfinite(gx1).all() and float(gw1.abs().max()) > 0


@pytest.mark.parametrize("drop", [None, (0.2, 0.1)])
def test_maxpool_backward_raises_the_scale_word(device, drop):
    """The three max-pool backward entries with ``din.amax``: the word ends at max |din| (= the largest masked pooled gradient) and the tensor carries the tag
    ``grad_operand`` takes instead of an ``sf_amax`` pass - with and without the dropout masks, routed and recomputed forms."""
    from satflow_amd import kernels as K

    g = torch.Generator().manual_seed(23)
    n, h, w, c = 6, 8, 12, 32
    x = torch.randn(n, h, w, c, generator=g).to(device)
    gy = (torch.randn(n, h // 2, w // 2, c, generator=g) * 3e-7).to(device)
    d = None if drop is None else (drop[0], drop[1], (h // 2) * (w // 2) * c * 2, 1234, 5678)
    y, route = K.maxpool2_route_fwd(x, None, None, d)
    for gx in (K.maxpool2_route_bwd(route, gy, tuple(x.shape), torch.float32, None, d), K.maxpool2_bwd(x, gy, None, d)):
        tag = getattr(gx, "_sf_amax", None)
        assert tag is not None and tag[1] == gx._version and tag[2] == gx.data_ptr()
        assert float(tag[0]) == float(gx.abs().max()) > 0
        op = K.grad_operand(gx)
        assert op.amax == tag[0].data_ptr()
        gx.mul_(2.0)   # written in place: the tag is void, grad_operand measures again
        op2 = K.grad_operand(gx)
        assert op2.amax != tag[0].data_ptr() and float(op2._keep) == float(gx.abs().max())


@pytest.mark.parametrize("n,cin,cout,h,w", [(3, 32, 64, 32, 32), (2, 160, 256, 40, 24), (2, 48, 96, 33, 17)])
def test_convolution_statistics_epilogue(device, n, cin, cout, h, w):
    """sf_conv3x3_fwd_stats in "f32e" (what saves the BatchNorm behind conv2 / conv3 its statistics pass): the output is the plain launch's bit for bit, the
    per-tile sums and sums of squares are those of the stored fp32 values (float64 reference over 32 x 16 tiles, ragged tiles included)."""
    from satflow_amd.functional import ConvEngine, conv3x3

    g = torch.Generator().manual_seed(n * 7 + cin)
    x = torch.randn(n, h, w, cin, generator=g).to(device)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).to(device)
    b = torch.randn(cout, generator=g).to(device)
    eng = ConvEngine([cin], cout)
    with torch.no_grad():
        y0 = conv3x3(eng, x, wt, b)
        y1, st = conv3x3(eng, x, wt, b, want_stats=True)
    assert st is not None and torch.equal(y0[..., :cout], y1[..., :cout])
    tx, ty = (w + 15) // 16, (h + 31) // 32
    assert st.tiles == tx * ty
    yd = y1[..., :cout].double()
    ref = torch.zeros(n, ty, tx, cout, 2, dtype=torch.float64, device=device)
    for j in range(ty):
        for i in range(tx):
            blk = yd[:, 32 * j:32 * j + 32, 16 * i:16 * i + 16]
            ref[:, j, i, :, 0] = blk.sum((1, 2))
            ref[:, j, i, :, 1] = (blk * blk).sum((1, 2))
    got = st.data.view(n, ty, tx, st.np, 2)[..., :cout, :].double()
    assert torch.isfinite(got).all()
    err = (got - ref).abs().max() / ref[..., 1].abs().max().clamp_min(1.0)
    assert float(err) < 1e-5, float(err)


def test_convgru_steps_repeat_bit_for_bit_after_idle_gaps(device):
    """The per-step ConvGRU forward (24 steps over 24 maps of 16 x 16, hidden 64: MetNet's recurrent part in the fp32 modes) gives the SAME bits every time it
    is run, also when the GPU idled in between.  Round 6: in "f32e" it did not - the chunk loop's barrier was reached with the wave's last ds_write of the
    staged input still in the LDS queue (hipcc had dropped that wait from __syncthreads() in the SF_SPLIT3 build), and after an idle gap a wave of another SIMD
    read one 16-byte piece early in every third repetition: errors of 1e-6 here, 1e-3 on MetNet's final state, a parity failure once in ~25 runs.  The kernels
    wait for their LDS writes explicitly now (conv3x3_bf16.hip); tests/test_host_cpu.py checks the ISA for the pattern."""
    import time

    from satflow_amd.models import MetNet

    torch.manual_seed(1234)
    net = MetNet(input_channels=12, sat_channels=12, input_size=64, output_channels=12, hidden_dim=64, forecast_steps=12).to(device).train()
    rnn = net.temporal_enc.rnn
    feat = torch.randn(24 * 24, 16, 16, 256, generator=torch.Generator().manual_seed(7)).to(device)
    ref = None
    for it in range(30):
        torch.cuda.synchronize()
        time.sleep(0.25)
        with torch.no_grad():
            _, last = rnn.run(feat, 24, 24, input_dropout_done=True)
        torch.cuda.synchronize()
        cur = last[-1].clone()
        if ref is None:
            ref = cur
        else:
            assert torch.equal(cur, ref), f"repetition {it}: max |difference| {float((cur - ref).abs().max()):.3e}"
