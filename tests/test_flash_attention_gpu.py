"""sf_flash_attention_fwd / _bwd (the fused self-attention of the DGMR / DVD-GAN discriminators in the 16-bit compute modes; reference
satflow/models/layers/Discriminator.py:104-126) against float64 evaluations and against the materialised three-step path it replaces."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.fixture
def mode16(request):
    import satflow_amd
    satflow_amd.set_compute_dtype(request.param)
    yield request.param
    satflow_amd.set_compute_dtype("f32")


@pytest.mark.parametrize("mode16", ["bf16", "f16"], indirect=True)
@pytest.mark.parametrize("b,n,dqk,dv,scale", [(2, 256, 32, 256, 1.0), (3, 384, 16, 128, 0.5), (1, 128, 32, 32, 1.0), (2, 1024, 32, 64, 0.25), (1, 4096, 32, 256, 1.0)])
def test_fused_attention_against_float64_of_rounded_operands(device, mode16, b, n, dqk, dv, scale):
    """Output and log-sum-exp against float64 on operands rounded to the mode's 16-bit type (the kernel rounds q * scale, k, v and the
    un-normalised probabilities: 3e-3 / 5e-4 on the output), gradients against float64 autograd of the un-rounded function (16-bit operand
    rounding on every product of the chain: 3e-2 at 8 mantissa bits)."""
    from satflow_amd import kernels as K

    dt = torch.bfloat16 if mode16 == "bf16" else torch.float16
    g = torch.Generator().manual_seed(b * 1000 + n + dv)
    q, k = (torch.randn(b, n, dqk, generator=g).to(device) for _ in range(2))
    v, dout = (torch.randn(b, n, dv, generator=g).to(device) for _ in range(2))
    assert K.flash_attention_ok(q, k, v)
    out, lse = K.flash_attention_fwd(q, k, v, scale)
    r = lambda t: t.to(dt).double()
    S = r(q * scale) @ r(k).transpose(1, 2)
    ref = torch.softmax(S, -1) @ r(v)
    assert _rel(out, ref) < (3e-3 if mode16 == "bf16" else 5e-4), _rel(out, ref)   # (the un-normalised probabilities are rounded to 8 / 11 bits)
    assert float((lse.double() - torch.logsumexp(S, -1)).abs().max()) < 1e-4
    dq, dk, dvg = K.flash_attention_bwd(q, k, v, out, lse, dout, scale)
    qd, kd, vd = (t.double().requires_grad_() for t in (q, k, v))
    (torch.softmax(scale * qd @ kd.transpose(1, 2), -1) @ vd).backward(dout.double())
    tol = 3e-2 if mode16 == "bf16" else 5e-3
    for name, got, want in (("dq", dq, qd.grad), ("dk", dk, kd.grad), ("dv", dvg, vd.grad)):
        assert torch.isfinite(got).all(), name
        assert _rel(got, want) < tol, (name, _rel(got, want))


@pytest.mark.parametrize("mode16", ["bf16", "f16"], indirect=True)
def test_fused_attention_equals_the_materialised_path(device, mode16):
    """functional_gan.attention: the fused kernel against bmm -> softmax -> bmm with the same 16-bit operand rounding (outputs and all three
    gradients; the two differ in WHERE the probabilities are rounded - before / after normalisation - and in the summation order)."""
    from satflow_amd import functional_gan as FG

    g = torch.Generator().manual_seed(11)
    n, hw, dqk, dv = 3, 1024, 32, 128
    q0, k0 = (torch.randn(n, hw, dqk, generator=g).to(device) for _ in range(2))
    v0, cot = (torch.randn(n, hw, dv, generator=g).to(device) for _ in range(2))

    def run(fused):
        q, k, v = (t.clone().requires_grad_() for t in (q0, k0, v0))
        out = FG.attention(q, k, v) if fused else FG.bmm(FG.softmax_last(FG.bmm(q, k.transpose(1, 2), lowp=True)), v, lowp=True)
        out.backward(cot)
        return out.detach(), q.grad, k.grad, v.grad

    a, b = run(True), run(False)
    tol = 2e-2 if mode16 == "bf16" else 3e-3
    for name, x, y in zip(("out", "dq", "dk", "dv"), a, b):
        assert _rel(x, y) < tol, (name, _rel(x, y))


def test_fused_attention_is_not_taken_in_fp32_mode(device):
    """The parity mode keeps the exact-fp32 materialised path (the fused kernel has no fp32 build and says so)."""
    import satflow_amd
    from satflow_amd import kernels as K
    from satflow_amd._hip import lib

    satflow_amd.set_compute_dtype("f32")
    q = torch.randn(1, 128, 32, device=device)
    v = torch.randn(1, 128, 64, device=device)
    assert not K.flash_attention_ok(q, q, v)
    out = torch.empty_like(v)
    rc = lib().sf_flash_attention_fwd(q.data_ptr(), 32, q.data_ptr(), 32, v.data_ptr(), 64, 1, 128, 32, 64, 1.0, out.data_ptr(), 64, None, 0, None)
    assert rc != 0 and b"16-bit" in lib().sf_last_error_string()
