"""GPU: the persistent ConvGRU sequence kernel (sf_convgru_seq_fwd: all timesteps in one launch, hidden state resident in
registers / LDS) against the per-step kernel it replaces (sf_convgru_step_fwd, itself checked against the oracle in
tests/test_metnet_gpu.py / test_bf16_gpu.py).  Same K order, same bf16 rounding of the state, same epilogue formulas: the
states and the saved gates must agree BIT FOR BIT, for fp32-stored and (widened) bf16-stored x-parts, with and without an
initial state, on full 16x16 maps and on ragged ones."""
import pytest
import torch

import satflow_amd

pytestmark = pytest.mark.gpu


@pytest.fixture()
def bf16_mode():
    satflow_amd.set_compute_dtype("bf16")
    yield
    satflow_amd.set_compute_dtype("f32")


@pytest.mark.parametrize("Tn,n,H,W,hid", [(5, 3, 16, 16, 64), (4, 2, 16, 16, 32), (3, 2, 12, 10, 24), (3, 1, 5, 7, 16), (24, 4, 16, 16, 64), (1, 1, 16, 16, 64),
                                          (24, 96, 16, 16, 64), (6, 5, 11, 13, 48), (7, 128, 9, 16, 64), (3, 130, 16, 16, 64)])  # split over two workgroups: MetNet's size, ragged, 256 workgroups; one-workgroup kernel beyond the CU count
@pytest.mark.parametrize("gx_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("with_h0", [False, True])
def test_persistent_sequence_matches_per_step_kernel(device, bf16_mode, Tn, n, H, W, hid, gx_dtype, with_h0):
    from satflow_amd import kernels as K
    from satflow_amd._hip import T, cpad
    from satflow_amd.functional import GRUEngine

    g = torch.Generator().manual_seed(Tn * 1000 + H * 10 + hid)
    hidp = cpad(hid)
    eng = GRUEngine(16, hid)
    Wh = (torch.randn(3 * hid, hid, 3, 3, generator=g) * (0.6 / hid**0.5)).to(device)
    bh = torch.cat((torch.zeros(2 * hid), torch.randn(hid, generator=g) * 0.3)).to(device)
    packed, bp = K.pack_weights(Wh, bh, eng.h_fwd, False)
    gx = torch.randn(Tn * n, H, W, 3 * hidp, generator=g).to(device).to(gx_dtype)
    h0 = (torch.randn(n, H, W, hidp, generator=g) * 0.5).to(device) if with_h0 else None
    if h0 is not None:
        h0[..., hid:] = 0
    assert K.convgru_seq_supported(H, W, hidp)
    for gates_dtype in (torch.float32, torch.bfloat16):
        hs = torch.full((Tn, n, H, W, hidp), float("nan"), device=device)
        gates = torch.full((Tn, n, H, W, 4 * hidp), float("nan"), device=device).to(gates_dtype)
        ws = K.convgru_seq_fwd(gx, h0, Tn, n, H, W, packed, bp, hidp, hs, gates)
        if ws is not None:  # the two-workgroups-per-map kernel ran (H > 8, hidp > 32): no receiver gave up on its partner's boundary row
            torch.cuda.synchronize()
            assert int(ws[0]) == 0, "split kernel: a boundary-row hand-off timed out"
        # reference: one launch per step on the fp32 widening of the same x-part
        gxf = gx.float().view(Tn, n, H, W, 3 * hidp)
        hs_ref = torch.zeros(Tn, n, H, W, hidp, device=device)
        gates_ref = torch.zeros(Tn, n, H, W, 4 * hidp, device=device).to(gates_dtype)
        for t in range(Tn):
            K.convgru_step_fwd(T(gxf[t]), hs_ref[t - 1] if t else h0, n, H, W, packed, bp, hidp, hs_ref[t], gates_ref[t])
        torch.cuda.synchronize()
        assert torch.equal(hs[..., :hid], hs_ref[..., :hid]), f"states differ: max {float((hs[..., :hid] - hs_ref[..., :hid]).abs().max()):.3e}"
        gsel = torch.cat([torch.arange(q * hidp, q * hidp + hid) for q in range(4)]).to(device)
        assert torch.equal(gates[..., gsel].float(), gates_ref[..., gsel].float()), "saved gates differ"
        assert float(hs[..., :hid].abs().max()) > 0.05 and torch.isfinite(hs[..., :hid]).all()
    # without a gates tensor (inference): same states
    hs2 = torch.empty_like(hs)
    K.convgru_seq_fwd(gx, h0, Tn, n, H, W, packed, bp, hidp, hs2, None)
    assert torch.equal(hs2[..., :hid], hs_ref[..., :hid])


def test_convgru_module_uses_persistent_kernel_and_matches_per_step(device, monkeypatch):
    """ConvGRU.run in the bf16 modes: persistent forward == per-step forward (SF_GRU_PER_STEP=1), states and all gradients
    (the backward pass consumes the same saved tensors either way; in "bf16" mode the x-part is fp32 on both paths)."""
    from satflow_amd import functional as F
    from satflow_amd.models.metnet import ConvGRU

    B, T, cin, hid, h, w = 3, 6, 32, 64, 16, 16
    torch.manual_seed(1)
    rnn = ConvGRU(cin, hid, (3, 3), 1).to(device).eval()
    x = torch.randn(B, T, cin, h, w, generator=torch.Generator().manual_seed(2)).to(device)
    cot = torch.randn(B, hid, h, w, generator=torch.Generator().manual_seed(3)).to(device)
    res = {}
    satflow_amd.set_compute_dtype("bf16")
    try:
        for mode in ("persistent", "per_step"):
            if mode == "per_step":
                monkeypatch.setenv("SF_GRU_PER_STEP", "1")
            rnn.zero_grad()
            xd = x.clone().requires_grad_()
            xs = F._ToNHWC.apply(xd, B, T, cin, h, w, (T * cin * h * w, cin * h * w, h * w))
            seq, last = rnn.run(xs, T, B)
            out = F.nhwc_to_nchw(last[-1], hid)
            (out * cot).sum().backward()
            res[mode] = (out.detach().clone(), xd.grad.clone(), {k: p.grad.clone() for k, p in rnn.named_parameters()})
    finally:
        satflow_amd.set_compute_dtype("f32")
    assert torch.equal(res["persistent"][0], res["per_step"][0])
    assert torch.equal(res["persistent"][1], res["per_step"][1])
    for k in res["per_step"][2]:
        assert torch.equal(res["persistent"][2][k], res["per_step"][2][k]), k


@pytest.mark.parametrize("Tn,n,H,W,hid", [(5, 3, 16, 16, 64), (4, 2, 16, 16, 32), (3, 2, 12, 10, 64), (3, 1, 5, 7, 32), (24, 4, 16, 16, 64), (1, 2, 16, 16, 64),
                                          (24, 96, 16, 16, 64), (6, 128, 9, 13, 64), (3, 130, 16, 16, 64)])  # MetNet's size / ragged on 256 workgroups (split); beyond the CU count (one workgroup per map)
@pytest.mark.parametrize("use_seq,use_last", [(False, True), (True, False), (True, True)])
def test_persistent_backward_matches_per_step_kernels(device, bf16_mode, Tn, n, H, W, hid, use_seq, use_last):
    """sf_convgru_seq_bwd (the whole backward time loop in one launch) against sf_convgru_bwd_gates + sf_conv3x3_fwd per step: the
    bf16-stored dgx / dgh must agree bit for bit (same gate arithmetic, same K order, same summation order of the carried gradient)."""
    from satflow_amd import kernels as K
    from satflow_amd._hip import NULL, T, cpad
    from satflow_amd.functional import GRUEngine

    g = torch.Generator().manual_seed(Tn * 1000 + H * 10 + hid + 7)
    hidp = cpad(hid)
    eng = GRUEngine(16, hid)
    Wh = (torch.randn(3 * hid, hid, 3, 3, generator=g) * (0.6 / hid**0.5)).to(device)
    packed_t = K.pack_weights(Wh, None, eng.h_bwd, True)[0]
    gates = torch.rand(Tn, n, H, W, 4 * hidp, generator=g)
    gates[..., 2 * hidp:] = gates[..., 2 * hidp:] * 2 - 1  # candidate in (-1, 1), h2 any sign
    gates = gates.to(device).to(torch.bfloat16)
    hs = (torch.randn(Tn, n, H, W, hidp, generator=g) * 0.5).to(device)
    g_seq = torch.randn(Tn, n, H, W, hidp, generator=g).to(device) if use_seq else None
    g_last = torch.randn(n, H, W, hidp, generator=g).to(device) if use_last else None
    assert K.convgru_seq_bwd_supported(H, W, hidp, gates)
    dgx = torch.full((Tn, n, H, W, 3 * hidp), float("nan"), device=device).to(torch.bfloat16)
    dgh = torch.full((Tn, n, H, W, 3 * hidp), float("nan"), device=device).to(torch.bfloat16)
    ws = K.convgru_seq_bwd(g_seq, g_last, gates, hs, Tn, n, H, W, packed_t, hidp, dgx, dgh)
    if ws is not None:  # the two-workgroups-per-map kernel ran: no receiver gave up on its partner's boundary row
        torch.cuda.synchronize()
        assert int(ws[0]) == 0, "split kernel: a boundary-row hand-off timed out"
    # reference: the per-step kernels
    rgx, rgh = torch.zeros_like(dgx), torch.zeros_like(dgh)
    direct = torch.empty(n, H, W, hidp, device=device)
    carry = torch.empty(n, H, W, hidp, device=device)
    have = False
    for t in range(Tn - 1, -1, -1):
        src = []
        if g_seq is not None:
            src.append(T(g_seq[t]))
        if t == Tn - 1 and g_last is not None:
            src.append(T(g_last))
        if have:
            src += [T(direct), T(carry)]
        K.convgru_bwd_gates(src, gates[t], hs[t - 1] if t else None, hidp, rgx[t], rgh[t], direct if t else None)
        if t:
            K.conv3x3(T(rgh[t]), NULL, n, H, W, packed_t, None, eng.h_bwd, T(carry))
            have = True
    torch.cuda.synchronize()
    assert torch.isfinite(dgx.float()).all() and torch.isfinite(dgh.float()).all()
    for name, a, b in (("dgx", dgx, rgx), ("dgh", dgh, rgh)):
        for t in range(Tn - 1, -1, -1):  # report the first step (from the end) that differs
            assert torch.equal(a[t], b[t]), f"{name}[{t}] differs: max {float((a[t].float() - b[t].float()).abs().max()):.3e}"
    assert float(dgx.float().abs().max()) > 1e-3


class _RoundOperand(torch.autograd.Function):
    """bf16 rounding of a convolution operand with a straight-through gradient that is ALSO rounded to bf16 where the kernels round
    it: the cotangent of a convolution input is formed from bf16(dout) x bf16(W), which the oracle's conv backward reproduces by
    itself once its operands are rounded - so the pass-through here is the identity."""

    @staticmethod
    def forward(ctx, t):
        return t.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


@pytest.mark.parametrize("mode", ["bf16", "bf16a"])
@pytest.mark.parametrize("B,T,cin,hid,h,w", [(3, 6, 32, 64, 16, 16), (2, 5, 16, 32, 12, 10)])
def test_persistent_sequence_kernels_against_oracle(device, mode, B, T, cin, hid, h, w):
    """DIRECT oracle check of sf_convgru_seq_fwd / sf_convgru_seq_bwd (not via the per-step kernels): the ConvGRU module in the
    bf16 modes against oracle.metnet.convgru whose convolution operands (inputs, states, weights) are rounded to bf16 - everything
    else fp32 on both sides.  "bf16": persistent forward, per-step backward; "bf16a": both persistent, x-part / gates / gate
    gradients stored as bf16 (an extra 2^-9 relative rounding per stored value, hence the looser bounds)."""
    from conftest import rel_l2
    from oracle import metnet as M
    from satflow_amd import functional as F
    from satflow_amd import kernels as K
    from satflow_amd._hip import cpad
    from satflow_amd.models.metnet import ConvGRU

    torch.manual_seed(11)
    rnn = ConvGRU(cin, hid, (3, 3), 1).eval()
    with torch.no_grad():  # hot weights: gates leave the linear regime, the carried state matters
        for p_ in rnn.parameters():
            if p_.dim() > 1:
                p_.mul_(3.0)
            else:
                p_.copy_(torch.randn(p_.shape) * 0.3)
    P = {f"rnn.{k}": v.detach().clone().requires_grad_() for k, v in rnn.state_dict().items()}
    x = torch.randn(B, T, cin, h, w, generator=torch.Generator().manual_seed(2))
    cot_seq = torch.randn(B, T, hid, h, w, generator=torch.Generator().manual_seed(3))
    cot_last = torch.randn(B, hid, h, w, generator=torch.Generator().manual_seed(4))
    xr = x.clone().requires_grad_()
    ref_seq, ref_last = M.convgru(xr, P, "rnn", 1, operand=_RoundOperand.apply)
    ((ref_seq * cot_seq).sum() + (ref_last[-1] * cot_last).sum()).backward()

    rnn = rnn.to(device)
    satflow_amd.set_compute_dtype(mode)
    try:
        assert K.convgru_seq_supported(h, w, cpad(hid))
        xd = x.to(device).requires_grad_()
        xs = F._ToNHWC.apply(xd, B, T, cin, h, w, (T * cin * h * w, cin * h * w, h * w))
        seq, last = rnn.run(xs, T, B)
        out_seq = F._FromNHWC.apply(seq, (B, T, hid, h, w), B, T, hid, h, w, (T * hid * h * w, hid * h * w, h * w))
        out_last = F.nhwc_to_nchw(last[-1], hid)
        ((out_seq * cot_seq.to(device)).sum() + (out_last * cot_last.to(device)).sum()).backward()
    finally:
        satflow_amd.set_compute_dtype("f32")
    tight = mode == "bf16"
    e_seq, e_last = rel_l2(out_seq, ref_seq), rel_l2(out_last, ref_last[-1])
    print(f"   {mode}: states rel L2 {e_seq:.2e}, last {e_last:.2e}")
    assert e_seq < (1e-3 if tight else 1e-2) and e_last < (1e-3 if tight else 1e-2), (e_seq, e_last)
    e_dx = rel_l2(xd.grad, xr.grad)
    print(f"   {mode}: dx rel L2 {e_dx:.2e}")
    assert e_dx < (6e-3 if tight else 3e-2), e_dx
    for k, p_ in rnn.named_parameters():
        e = rel_l2(p_.grad, P[f"rnn.{k}"].grad)
        print(f"   {mode}: d{k} rel L2 {e:.2e}")
        assert e < (6e-3 if tight else 3e-2), (k, e)


# ----------------------------------------------------------------------------------------------------------------------
# The two-workgroups-per-map kernels when the launch does NOT own the GPU: partners are assigned by
# start-order tickets, so nothing depends on every workgroup being resident.  A parked kernel holds most CUs on a second
# stream (what an RCCL kernel under FlatAdam(overlap=True), another process or a CU mask do) while the sequence kernels run.
# ----------------------------------------------------------------------------------------------------------------------
def _seq_problem(device, Tn, n, H, W, hid, seed=5):
    from satflow_amd import kernels as K
    from satflow_amd._hip import cpad
    from satflow_amd.functional import GRUEngine

    g = torch.Generator().manual_seed(seed)
    hidp = cpad(hid)
    eng = GRUEngine(16, hid)
    Wh = (torch.randn(3 * hid, hid, 3, 3, generator=g) * (0.6 / hid**0.5)).to(device)
    bh = torch.cat((torch.zeros(2 * hid), torch.randn(hid, generator=g) * 0.3)).to(device)
    packed, bp = K.pack_weights(Wh, bh, eng.h_fwd, False)
    packed_t = K.pack_weights(Wh, None, eng.h_bwd, True)[0]
    gx = torch.randn(Tn * n, H, W, 3 * hidp, generator=g).to(device).to(torch.bfloat16)
    g_seq = torch.randn(Tn, n, H, W, hidp, generator=g).to(device)
    return dict(eng=eng, hidp=hidp, packed=packed, bp=bp, packed_t=packed_t, gx=gx, g_seq=g_seq)


def _run_seq(device, pr, Tn, n, H, W):
    from satflow_amd import kernels as K

    hidp = pr["hidp"]
    hs = torch.full((Tn, n, H, W, hidp), float("nan"), device=device)
    gates = torch.full((Tn, n, H, W, 4 * hidp), float("nan"), device=device).to(torch.bfloat16)
    dgx = torch.full((Tn, n, H, W, 3 * hidp), float("nan"), device=device).to(torch.bfloat16)
    dgh = torch.full_like(dgx, float("nan"))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    wf = K.convgru_seq_fwd(pr["gx"], None, Tn, n, H, W, pr["packed"], pr["bp"], hidp, hs, gates)
    wb = K.convgru_seq_bwd(pr["g_seq"], None, gates, hs, Tn, n, H, W, pr["packed_t"], hidp, dgx, dgh)
    e1.record()
    return hs, gates, dgx, dgh, wf, wb, (e0, e1)


@pytest.mark.parametrize("parked,n", [(96, 96), (200, 96), (250, 96), (200, 128)])
def test_split_kernels_with_most_cus_held_by_another_stream(device, bf16_mode, parked, n):
    """n = 96 / 128 maps (192 / 256 workgroups of 152 KB LDS: one per CU) while `parked` CUs are held by a kernel on another
    stream for 20 ms: results bit-identical to the undisturbed run, error words clear, and the sequence kernels really ran
    NEXT TO the parked kernel (they finish long before it does unless fewer than ~8 CUs are left)."""
    from tests import native

    Tn, H, W, hid = 24, 16, 16, 64
    pr = _seq_problem(device, Tn, n, H, W, hid)
    satflow_amd.clear_device_errors()
    ref = _run_seq(device, pr, Tn, n, H, W)
    torch.cuda.synchronize()
    assert ref[4] is not None and ref[5] is not None, "the two-workgroup kernels did not run"
    assert not satflow_amd.device_errors()
    t_free = ref[6][0].elapsed_time(ref[6][1])

    side = torch.cuda.Stream(device)
    census = torch.zeros(parked, dtype=torch.int32, device=device)
    torch.cuda.synchronize()
    park_us = 20000
    p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        p0.record()
        native.occupy_cus(parked, 96 * 1024, park_us, side, census)
        p1.record()
    import time
    time.sleep(0.002)  # the parked workgroups are on their CUs before the sequence kernels are enqueued
    got = _run_seq(device, pr, Tn, n, H, W)
    torch.cuda.synchronize()
    t_busy, t_park = got[6][0].elapsed_time(got[6][1]), p0.elapsed_time(p1)
    cus_held = len(set(int(v) >> 6 for v in census.tolist()))  # HW_ID without the wave / SIMD bits, per XCC
    print(f"   parked {parked} workgroups on {cus_held} CUs for {t_park:.1f} ms; sequence kernels {t_free:.2f} ms alone, {t_busy:.2f} ms beside them")
    assert not satflow_amd.device_errors(), satflow_amd.device_errors()
    for name, a, b in zip(("hs", "gates", "dgx", "dgh"), got[:4], ref[:4]):
        assert torch.isfinite(a.float()).all(), name
        assert torch.equal(a, b), f"{name} differs next to a parked kernel"
    assert t_park >= 0.9 * park_us / 1000
    if parked <= 200:  # >= 56 CUs left: the sequence kernels complete while the other kernel is still parked
        assert t_busy < 0.75 * t_park, (t_busy, t_park)


def test_split_kernel_failed_handoff_is_loud(device, bf16_mode):
    """A boundary row that never arrives (test hook: half 1 never sends; spin bound lowered to milliseconds): the kernel does not
    hang, sets the sticky error word, and every state / gradient of the starved workgroups is NaN - the product path
    (`check_device_errors`, called by FlatAdam.step) raises.
    The hook is NOT in the product library (no process-wide state there, ABI 6): this test re-runs itself in a child process whose
    SATFLOW_HIP_LIB is tests/native/libsatflow_hip_hooks.so - the product objects with convgru_seq.hip rebuilt with -DSF_TEST_HOOKS."""
    import ctypes as C
    import os
    import subprocess
    import sys

    from satflow_amd._hip import lib

    if not os.environ.get("SF_TEST_HOOKS_CHILD"):
        from tests import native

        assert not hasattr(lib(), "sf_convgru_seq_debug"), "the product library must not export the test hook"
        env = dict(os.environ, SATFLOW_HIP_LIB=native.build_hooks(), SF_TEST_HOOKS_CHILD="1")
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-s", f"{__file__}::test_split_kernel_failed_handoff_is_loud"],
                           cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
        return
    debug = lib().sf_convgru_seq_debug
    debug.restype, debug.argtypes = None, [C.c_int32, C.c_int32]

    Tn, n, H, W, hid = 6, 8, 16, 16, 64
    pr = _seq_problem(device, Tn, n, H, W, hid)
    satflow_amd.clear_device_errors()
    good = _run_seq(device, pr, Tn, n, H, W)
    torch.cuda.synchronize()
    assert not satflow_amd.device_errors()
    debug(2000, 1)
    try:
        bad = _run_seq(device, pr, Tn, n, H, W)
        torch.cuda.synchronize()
    finally:
        debug(0, -1)
    errs = satflow_amd.device_errors()
    assert len(errs) == 2 and all(k[0].startswith("convgru_seq_") for k in errs), errs
    hs, dgx = bad[0], bad[2]
    assert torch.isnan(hs[-1, :, :8]).all(), "the starved halves' last states must be NaN"   # rows 0..7 = half 0, which never received
    assert torch.isnan(hs[-1]).any(dim=(1, 2, 3)).all(), "every map must carry the failure"
    assert torch.isnan(dgx.float()[0]).any(dim=(1, 2, 3)).all()
    with pytest.raises(RuntimeError, match="hand-off"):
        satflow_amd.check_device_errors()
    satflow_amd.clear_device_errors()
    again = _run_seq(device, pr, Tn, n, H, W)
    torch.cuda.synchronize()
    assert not satflow_amd.device_errors()
    for a, b in zip(again[:4], good[:4]):
        assert torch.equal(a, b)
