"""The gradient sink (satflow_amd.functional.GradSink): weight-gradient kernels write a parameter's gradient straight into the optimizer's flat
buffer instead of returning a tensor for autograd's AccumulateGrad to add.  The host logic on the CPU; on the GPU the gradients of whole training
steps with the sink against the same steps with every gradient going through autograd (SF_NO_GRAD_SINK semantics): bit-identical.
"""
import pytest
import torch

from satflow_amd import functional as F


class _Owner:
    pass


class _Double(torch.autograd.Function):
    """y = 2 w (+ x): backward asks the sink where w's gradient goes, like the weight-gradient Functions do."""

    @staticmethod
    def forward(ctx, w, log):
        ctx.save_for_backward(w)
        ctx.log = log
        return 2.0 * w

    @staticmethod
    def backward(ctx, g):
        (w,) = ctx.saved_tensors
        out, ret = F.grad_out(w)
        out.copy_(2.0 * g)
        ctx.log.append(ret is None)
        return ret, None


def _fresh(n=2):
    sink = F.GradSink()
    sink.off = False
    owner = _Owner()
    flat_p, flat_g = torch.arange(8.0), torch.zeros(8)
    params = []
    for i in range(n):
        p = torch.nn.Parameter(torch.zeros(4))
        p.data = flat_p[4 * i: 4 * i + 4]
        p.grad = flat_g[4 * i: 4 * i + 4]
        sink.register(owner, p, p.grad)
        params.append(p)
    return sink, owner, flat_g, params


def test_sink_hands_a_destination_out_once_per_parameter_and_zero_grad(monkeypatch):
    sink, owner, flat_g, (p, q) = _fresh()
    monkeypatch.setattr(F, "GRAD_SINK", sink)
    log = []
    # two consumers of p in one graph: the first to run its backward writes in place, the second goes through autograd and is added on top
    (_Double.apply(p, log).sum() + 3.0 * _Double.apply(p, log).sum() + _Double.apply(q, log).sum()).backward()
    assert sorted(log) == [False, True, True]
    assert torch.equal(flat_g, torch.tensor([8.0] * 4 + [2.0] * 4))
    assert p.grad.data_ptr() == flat_g.data_ptr()          # still the flat view
    # a second backward pass without zero_grad (gradient accumulation): the sink is closed, autograd accumulates
    log.clear()
    _Double.apply(q, log).sum().backward()
    assert log == [False]
    assert torch.equal(flat_g[4:], torch.tensor([4.0] * 4))
    # zero_grad re-opens it
    flat_g.zero_()
    sink.reopen(owner)
    log.clear()
    _Double.apply(q, log).sum().backward()
    assert log == [True] and torch.equal(flat_g[4:], torch.tensor([2.0] * 4))


def test_sink_ignores_what_is_not_a_registered_parameter(monkeypatch):
    sink, owner, flat_g, (p, q) = _fresh()
    monkeypatch.setattr(F, "GRAD_SINK", sink)
    assert sink.dest(None) is None
    assert sink.dest(torch.zeros(4)) is None                                      # unknown tensor
    assert sink.dest(p[:2]) is None                                               # same address, a slice
    two = torch.nn.Parameter(torch.zeros(2, 2))
    two.data = p.data.view(2, 2)
    assert sink.dest(two.t()) is None                                             # same address and size, another element order
    log = []
    _Double.apply(p, log).sum().backward()                                        # (dest() arms an end-of-backward callback: call it inside one)
    assert log == [True]
    # an optimizer that is gone: its entries are dropped, not matched by whatever lives at the address next
    sink2, owner2, flat_g2, (p2, _) = _fresh()
    del owner2
    import gc

    gc.collect()
    assert not sink2.table and not sink2.taken, "the views of a collected optimizer's gradient buffer are released with it"
    assert sink2.dest(p2) is None
    # the A/B switch
    sink.reopen(owner)
    sink.off = True
    assert sink.dest(q) is None


@pytest.mark.gpu
@pytest.mark.parametrize("model,mode", [("convlstm", "bf16a"), ("metnet", "f32"), ("metnet", "bf16a")])
def test_training_step_gradients_with_and_without_the_sink(device, model, mode):
    """One optimizer step's gradients with the sink (the kernels write flat_g) against the same step with every gradient returned to autograd; a
    second backward pass without zero_grad doubles them (accumulation falls back to autograd); the parameters' .grad stay the flat views.
    The ConvLSTM step is deterministic: bit for bit.  MetNet's is not (double-precision atomics in its BatchNorm sums; two builds of the SAME
    configuration differ by ~1e-3 of a gradient's scale in bf16a, where an ulp flips a bf16 rounding now and then): per parameter within 1e-5 of the
    gradient's scale in f32 mode, 5e-2 in bf16a - a gradient written to the wrong place is off by its whole scale."""
    import satflow_amd
    from satflow_amd.optim import FlatAdam

    old_mode = satflow_amd.compute_dtype_name()
    satflow_amd.set_compute_dtype(mode)
    try:
        def build():
            torch.manual_seed(5)
            if model == "convlstm":
                from satflow_amd.models.conv_lstm import ConvLSTM

                net = ConvLSTM(4, 16, 4).to(device)
                x = torch.randn(2, 3, 4, 32, 32, device=device)
                run = lambda: net(x, forecast_steps=2).square().mean()
            else:
                from satflow_amd.models.metnet import MetNet

                net = MetNet(input_channels=4, sat_channels=4, input_size=32, output_channels=4, hidden_dim=16, forecast_steps=2, temporal_dropout=0.0).to(device)
                net.train()
                x = torch.randn(2, 3, 4, 128, 128, device=device)

                def run():
                    torch.manual_seed(11)  # the dropout seeds are drawn on the host
                    return net(x).square().mean()
            return net, run

        results = []
        for off in (False, True):
            F.GRAD_SINK.off = off
            net, run = build()
            opt = FlatAdam(net.parameters(), lr=1e-3)
            assert opt.direct_grads
            opt.zero_grad()
            run().backward()
            torch.cuda.synchronize()
            g1 = opt.flat_g.clone()
            taken = len(F.GRAD_SINK.taken[id(opt)])
            assert (taken > 0) == (not off)
            for p, o in zip(opt.params, opt.offsets):
                assert p.grad is not None and p.grad.data_ptr() == opt.flat_g.data_ptr() + 4 * o
            run().backward()                      # accumulation: nothing may be overwritten
            torch.cuda.synchronize()
            g2 = opt.flat_g.clone()
            opt.zero_grad()
            run().backward()
            torch.cuda.synchronize()
            results.append((g1, g2, opt.flat_g.clone(), taken, list(zip(opt.offsets, [p.numel() for p in opt.params], [n for n, _ in net.named_parameters()]))))
            F.GRAD_SINK.unregister(opt)
        (a1, a2, a3, taken, layout), (b1, b2, b3, _, _) = results
        assert taken >= 8, taken
        assert torch.isfinite(a1).all() and a1.abs().max() > 0
        if model == "convlstm":
            assert torch.equal(a1, b1) and torch.equal(a3, b3) and torch.equal(a1, a3)
            # second pass of the accumulation: the sum of two equal gradients (fp32: exactly twice) whichever way the first one was written
            assert torch.equal(a2, b2) and torch.equal(a2, 2.0 * a1)
        else:
            tol = 1e-5 if mode == "f32" else 5e-2
            for x_, y_, what in ((a1, b1, "first pass"), (a3, b3, "after zero_grad"), (a2, 2.0 * b1, "accumulated"), (b2, 2.0 * b1, "accumulated, no sink")):
                floor = 1e-2 * float(y_.abs().max())   # (a bias in front of a BatchNorm has a zero gradient: what is there is rounding noise)
                for off_, n_, name in layout:
                    scale = max(float(y_[off_:off_ + n_].abs().max()), floor)
                    err = float((x_[off_:off_ + n_] - y_[off_:off_ + n_]).abs().max())
                    assert err <= tol * max(scale, 1e-12), f"{what}: {name}: {err:.3e} against a scale of {scale:.3e}"
    finally:
        F.GRAD_SINK.off = bool(__import__("os").environ.get("SF_NO_GRAD_SINK"))
        satflow_amd.set_compute_dtype(old_mode)


@pytest.mark.gpu
@pytest.mark.parametrize("sink", [False, True])
def test_param_blocks_equal_cat_of_slices(device, sink):
    """``functional.param_blocks`` (sf_copy_blocks: one launch each way) against the torch.cat / slice form it replaces - the ConvGRU cell's regrouping
    of conv_zr / conv_h1 / conv_h2 into an x-part and an h-part: outputs and all six parameter gradients bit-identical (copies and zero fills only),
    with the gradients going through autograd and going straight into an optimizer's flat buffer."""
    from satflow_amd.models.metnet import ConvGRUCell
    from satflow_amd.optim import FlatAdam

    torch.manual_seed(3)
    cell = ConvGRUCell(24, 16).to(device)
    with torch.no_grad():
        for p in cell.parameters():
            p.copy_(torch.randn_like(p))
    cots = [torch.randn(48, 24, 3, 3, device=device), torch.randn(48, device=device), torch.randn(48, 16, 3, 3, device=device), torch.randn(48, device=device)]
    import os

    def grads(blocks: bool):
        os.environ.pop("SF_NO_PARAM_BLOCKS", None)
        if not blocks:
            os.environ["SF_NO_PARAM_BLOCKS"] = "1"
        try:
            for p in cell.parameters():
                p.grad = None
            opt = FlatAdam(cell.parameters(), lr=1e-3) if sink and blocks else None
            if opt is not None:
                opt.zero_grad()
            outs = cell.regrouped()
            sum((o * c).sum() for o, c in zip(outs, cots)).backward()
            torch.cuda.synchronize()
            taken = len(F.GRAD_SINK.taken[id(opt)]) if opt is not None else 0
            res = [o.detach().clone() for o in outs], [p.grad.detach().clone() for p in cell.parameters()], taken
            if opt is not None:
                F.GRAD_SINK.unregister(opt)
            return res
        finally:
            os.environ.pop("SF_NO_PARAM_BLOCKS", None)

    o_ref, g_ref, _ = grads(False)
    o_new, g_new, taken = grads(True)
    assert taken == (6 if sink else 0)
    for a, b in zip(o_new, o_ref):
        assert a.shape == b.shape and torch.equal(a, b)
    for a, b, (name, _) in zip(g_new, g_ref, cell.named_parameters()):
        assert torch.equal(a, b), name


def test_a_plain_autograd_pass_closes_the_owner_too(monkeypatch):
    """Advisor r5 (medium): a backward pass that reaches a registered parameter ONLY through plain autograd operations used to leave its owner open, and the
    next pass's sink-aware Function then OVERWROTE the accumulated gradient (2 instead of 5).  The tensor hook the sink puts on every registered parameter
    closes the owner at the end of any pass that produced a gradient for it."""
    sink, owner, flat_g, (p, q) = _fresh()
    monkeypatch.setattr(F, "GRAD_SINK", sink)
    (3.0 * p).sum().backward()
    assert torch.equal(flat_g[:4], torch.tensor([3.0] * 4))
    log = []
    _Double.apply(p, log).sum().backward()
    assert log == [False], "the second pass must go through autograd (accumulate), not overwrite"
    assert torch.equal(flat_g[:4], torch.tensor([5.0] * 4))
    # q was never reached by the first pass - but the owner (one optimizer, one zero_grad) is closed as a whole
    _Double.apply(q, log).sum().backward()
    assert log == [False, False] and torch.equal(flat_g[4:], torch.tensor([2.0] * 4))


def test_frozen_and_unrequested_parameters_get_no_destination(monkeypatch):
    """Advisor r5 (medium): a parameter frozen after the optimizer registered it must not receive a gradient in the flat buffer (the optimizer updates the
    whole buffer); neither must an input whose gradient autograd did not ask for (``needs_input_grad``)."""
    sink, owner, flat_g, (p, q) = _fresh()
    monkeypatch.setattr(F, "GRAD_SINK", sink)
    q.requires_grad_(False)
    assert sink.dest(q) is None
    out, ret = F.grad_out(q)
    assert ret is not None and out.data_ptr() != flat_g[4:].data_ptr()
    out, ret = F.grad_out(p, needed=False)
    assert ret is None and out.data_ptr() != flat_g.data_ptr() and torch.equal(flat_g, torch.zeros(8))
    with F.no_grad_sink():   # every gradient through autograd inside the block (for torch.autograd.grad on registered parameters)
        assert sink.off and sink.dest(p) is None
    assert not sink.off


def test_a_failed_backward_does_not_leave_the_sink_armed(monkeypatch):
    """Advisor r5 (low): the engine runs no final callbacks when a backward pass raises; ``reopen()`` (zero_grad) drops the stale pass."""
    sink, owner, flat_g, (p, q) = _fresh()
    monkeypatch.setattr(F, "GRAD_SINK", sink)

    class _Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 1.0

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError("boom")

    log = []
    with pytest.raises(RuntimeError, match="boom"):
        (_Boom.apply(_Double.apply(p, log)).sum() + _Double.apply(q, log).sum()).backward()
    flat_g.zero_()
    sink.reopen(owner)
    assert not sink._armed and not sink._pass_owners
    log.clear()
    _Double.apply(p, log).sum().backward()
    assert log == [True]
    _Double.apply(p, log).sum().backward()   # ... and that pass's end closed the owner again
    assert log == [True, False] and torch.equal(flat_g[:4], torch.tensor([4.0] * 4))
