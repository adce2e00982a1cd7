"""GPU parity of CloudGAN with the ConvLSTM generator against golden vectors captured from the reference itself
(tests/golden/make_golden.py::cloudgan_cases imports satflow.models.cloudgan / satflow.models.gan): the PatchGAN discriminator
alone, then both optimizer steps of `training_step` - losses, EVERY parameter gradient of generator and discriminator, and the
BatchNorm running statistics after the two steps (their update order follows the reference's per-timestep call order).
fp32, rtol 1e-4 / atol 1e-5.  Reference: satflow/models/cloudgan.py:121-189, gan/discriminators.py:70-223."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, assert_close

pytestmark = pytest.mark.gpu


def _load(name):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(GOLDEN, name)).items()}


def test_patch_discriminator_golden(device):
    from satflow_amd.models.gan import NLayerDiscriminator

    G = _load("cloudgan_discriminator.npz")
    D = NLayerDiscriminator(5, ndf=8, n_layers=3, norm_layer=torch.nn.BatchNorm2d)
    D.load_state_dict({k[len("param."):]: v for k, v in G.items() if k.startswith("param.")}, strict=False)
    D = D.to(device).train()
    x = G["x"].to(device).requires_grad_()
    out = D(x)
    assert_close(out, G["out"], "patch logits")
    (out * G["cot"].to(device)).sum().backward()
    assert_close(x.grad, G["dx"], "dx", grad=True, force_rel=True)
    for k, p in D.named_parameters():
        assert_close(p.grad, G[f"grad.{k}"], f"d{k}", grad=True)
    for k, b in D.named_buffers():  # running statistics after one training-mode call
        assert_close(b.float(), G[f"buffer.{k}"].float(), k)


@pytest.mark.parametrize("case", ["small", "rect"])
def test_cloudgan_training_steps_golden(device, case):
    from satflow_amd.models import CloudGAN

    G = _load(f"cloudgan_{case}.npz")
    B, T, C, H, W = G["images"].shape
    fs, lam, nf = int(G["forecast_steps"]), float(G["lambda_l1"]), int(G["num_filters"])
    m = CloudGAN(forecast_steps=fs, input_channels=C, num_filters=nf, generator_model="convlstm", norm="batch", discriminator_model="basic",
                 loss="vanilla", scheduler="cosine", lambda_l1=lam, channels_per_timestep=C, condition_time=True)
    m.generator.load_state_dict({k[len("gen."):]: v for k, v in G.items() if k.startswith("gen.")})
    m.discriminator.load_state_dict({k[len("disc."):]: v for k, v in G.items() if k.startswith("disc.")})
    m = m.to(device).train()
    batch = (G["images"].to(device), G["future"].to(device))
    for idx, tag in ((0, "g"), (1, "d")):  # same order as the golden run: the discriminator's running statistics accumulate
        m.zero_grad()
        out = m.training_step(batch, 0, idx)
        out["loss"].backward()
        assert_close(out["loss"], G[f"{tag}_loss"], f"{tag}_loss", rtol=1e-5, atol=1e-6)
        assert f"train/{tag}_loss" in m.logged
        for k, p in m.generator.named_parameters():
            ref = G[f"{tag}_grad.gen.{k}"]
            assert_close(p.grad if p.grad is not None else torch.zeros_like(p), ref, f"{tag} step d(gen.{k})", grad=True)
        for k, p in m.discriminator.named_parameters():
            ref = G[f"{tag}_grad.disc.{k}"]
            assert_close(p.grad if p.grad is not None else torch.zeros_like(p), ref, f"{tag} step d(disc.{k})", grad=True)
    assert {f"train/frame_{i}_l1_loss" for i in range(fs)} <= set(m.logged) and {f"train/frame_{i}_d_loss" for i in range(fs)} <= set(m.logged)
    sd = m.discriminator.state_dict()
    for k, v in G.items():
        if k.startswith("disc_after."):
            assert_close(sd[k[len("disc_after."):]].float(), v.float(), k)


def test_cloudgan_validation_step_and_eval(device):
    """validation_step (reference cloudgan.py:271-313) under model.eval(): metric names, finite losses, no statistics update."""
    from oracle import cloudgan as OC
    from satflow_amd.models import CloudGAN

    G = _load("cloudgan_small.npz")
    B, T, C, H, W = G["images"].shape
    fs = int(G["forecast_steps"])
    m = CloudGAN(forecast_steps=fs, input_channels=C, num_filters=int(G["num_filters"]), generator_model="convlstm", discriminator_model="basic",
                 lambda_l1=float(G["lambda_l1"]), channels_per_timestep=C, condition_time=True, scheduler="cosine")
    m.generator.load_state_dict({k[len("gen."):]: v for k, v in G.items() if k.startswith("gen.")})
    m.discriminator.load_state_dict({k[len("disc."):]: v for k, v in G.items() if k.startswith("disc.")})
    m = m.to(device).eval()
    before = {k: v.clone() for k, v in m.discriminator.state_dict().items() if "running" in k}
    with torch.no_grad():
        out = m.validation_step((G["images"].to(device), G["future"].to(device)), 0)
    assert {"val/d_loss", "val/g_loss", "val/loss"} <= set(m.logged) and {f"val/frame_{i}_l1_loss" for i in range(fs)} <= set(m.logged)
    assert torch.isfinite(out["val/discriminator_loss"]) and torch.isfinite(out["val/generator_loss"])
    assert_close(m.logged["val/loss"], out["val/discriminator_loss"] + out["val/generator_loss"], "val/loss", rtol=1e-6, atol=1e-7)
    for k, v in before.items():
        assert torch.equal(m.discriminator.state_dict()[k], v)
    # the L1 part does not depend on the discriminator: check it against the oracle's generator
    gen = {k[len("gen."):]: v for k, v in G.items() if k.startswith("gen.")}
    ref = OC.O.convlstm_forward(G["images"], fs, gen)
    l1 = sum(torch.nn.functional.l1_loss(ref[:, :, i], G["future"][:, i]) for i in range(fs)) / fs * float(G["lambda_l1"])
    got = sum(m.logged[f"val/frame_{i}_l1_loss"] for i in range(fs)) / fs
    assert_close(got, l1, "mean val l1", rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("n,cin,cout,h,w,k,stride,pad,slope", [(2, 5, 8, 20, 18, 4, 2, 1, 0.2), (1, 16, 32, 9, 11, 4, 1, 1, 1.0), (3, 12, 40, 16, 16, 3, 1, 1, 1.0),
                                                              (2, 8, 1, 7, 6, 4, 1, 1, 1.0), (1, 3, 7, 13, 17, 5, 3, 2, 0.1), (2, 24, 16, 8, 8, 1, 1, 0, 1.0)])
def test_generic_conv2d_vs_torch(device, n, cin, cout, h, w, k, stride, pad, slope):
    """sf_conv2d_fwd / bwd_data / bwd_weight (+ fused LeakyReLU) against torch CPU conv2d."""
    import torch.nn.functional as TF

    from satflow_amd import functional as F

    g = torch.Generator().manual_seed(n * 100 + cin + cout)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (k * cin**0.5)
    b = torch.randn(cout, generator=g)
    xr, wr, br = x.clone().requires_grad_(), wt.clone().requires_grad_(), b.clone().requires_grad_()
    ref = TF.leaky_relu(TF.conv2d(xr, wr, br, stride=stride, padding=pad), slope)
    cot = torch.randn(ref.shape, generator=g)
    (ref * cot).sum().backward()
    xd, wd, bd = (t.to(device).requires_grad_() for t in (x, wt, b))
    y = F.nhwc_to_nchw(F.conv2d(F.nchw_to_nhwc(xd), wd, bd, stride, pad, slope), cout)
    (y * cot.to(device)).sum().backward()
    assert_close(y, ref, "conv2d out")
    assert_close(xd.grad, xr.grad, "conv2d dx", grad=True)
    assert_close(wd.grad, wr.grad, "conv2d dW", grad=True)
    assert_close(bd.grad, br.grad, "conv2d db", grad=True)


# ---- the rest of the discriminator / objective surface (goldens: make_golden.py cloudgan, part 3) ---------------------------------
def _more(device):
    z = np.load(os.path.join(GOLDEN, "cloudgan_more.npz"))
    return {k: torch.from_numpy(np.asarray(z[k])).to(device) for k in z.files}


@pytest.mark.parametrize("tag", ["enhanced", "pixel", "instance"])
def test_discriminator_variants_match_reference(device, tag):
    """CloudGANDiscriminator ("enhanced": the reference constructor's default), PixelDiscriminator and the PatchGAN with InstanceNorm2d
    (reference gan/discriminators.py:139-312, gan/common.py:19-22): scores, input gradient, every parameter gradient."""
    import functools

    from satflow_amd.models.gan import CloudGANDiscriminator, NLayerDiscriminator, PixelDiscriminator

    g = _more(device)
    D = {"enhanced": lambda: CloudGANDiscriminator(input_channels=5, num_filters=8, num_stages=3),
         "pixel": lambda: PixelDiscriminator(5, ndf=8, norm_layer=torch.nn.BatchNorm2d),
         "instance": lambda: NLayerDiscriminator(5, ndf=8, n_layers=2, norm_layer=functools.partial(torch.nn.InstanceNorm2d, affine=False, track_running_stats=False))}[tag]()
    D = D.to(device).train()
    x = g[f"{tag}.x"].clone().requires_grad_()
    if tag == "enhanced":
        with torch.no_grad():
            D(x.detach())  # materialises the LazyLinear, as the reference's first call does
    D.load_state_dict({k[len(tag) + 7:]: v for k, v in g.items() if k.startswith(f"{tag}.param.")}, strict=False)
    out = D(x)
    assert_close(out, g[f"{tag}.out"], f"{tag}: scores")
    (out * g[f"{tag}.cot"]).sum().backward()
    assert_close(x.grad, g[f"{tag}.dx"], f"{tag}: dx", grad=True)
    for k, p in D.named_parameters():
        want = g[f"{tag}.grad.{k}"]
        if float(want.abs().max()) < 1e-4 and k.endswith("bias"):  # a bias in front of a normalisation layer: exact zero gradient
            assert float(p.grad.abs().max()) < 1e-3, k
        else:
            assert_close(p.grad, want, f"{tag}: d{k}", grad=True)


@pytest.mark.parametrize("mode", ["vanilla", "lsgan", "wgangp"])
def test_gan_loss_modes_match_reference(device, mode):
    from satflow_amd.models.gan import GANLoss

    g = _more(device)
    crit = GANLoss(mode).to(device)
    for real in (True, False):
        pr = g["loss.pred"].clone().requires_grad_()
        loss = crit(pr, real)
        assert_close(loss, g[f"loss.{mode}.{int(real)}"], f"{mode} real={real}")
        loss.backward()
        assert_close(pr.grad, g[f"loss.{mode}.{int(real)}.grad"], f"{mode} real={real}: gradient", grad=True)


def test_cloudgan_default_discriminator_and_objectives_run(device):
    """The reference constructor's defaults that used to raise: discriminator_model="enhanced", loss "lsgan" / "wgangp", norm="instance"
    (with the PatchGAN).  One generator and one discriminator step each: finite losses, gradients on the trained network only."""
    from satflow_amd.models import CloudGAN

    for kw in (dict(discriminator_model="enhanced", loss="lsgan"), dict(discriminator_model="enhanced", loss="wgangp"),
               dict(discriminator_model="basic", loss="vanilla", norm="instance"), dict(discriminator_model="pixel", loss="lsgan")):
        torch.manual_seed(3)
        m = CloudGAN(forecast_steps=2, input_channels=3, num_filters=8, generator_model="convlstm", channels_per_timestep=3, condition_time=True, **kw).to(device)
        x = torch.randn(2, 3, 3, 32, 32, device=device)
        y = torch.rand(2, 2, 3, 32, 32, device=device)
        for idx in (0, 1):
            m.zero_grad()
            loss = m.training_step((x, y), 0, idx)["loss"]
            assert torch.isfinite(loss), (kw, idx)
            loss.backward()
            net = m.generator if idx == 0 else m.discriminator
            assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters()), (kw, idx)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("cin,cout,n,h,w,stride", [(12, 64, 2, 32, 32, 2), (64, 128, 3, 16, 24, 2), (5, 7, 1, 6, 10, 2), (128, 256, 2, 16, 16, 1), (256, 1, 2, 15, 15, 1), (20, 33, 1, 7, 9, 1)])
def test_conv4x4_on_the_3x3_kernels(device, mode, cin, cout, n, h, w, stride):
    """PatchGAN's ``Conv2d(k=4, stride 1|2, padding=1)`` as ONE 3x3 MFMA convolution (``functional_gan.conv4x4_as_3x3``: 2x2 pixel blocks folded into
    channels / zero-extended 5x5 kernel) against ``torch.nn.functional.conv2d``: output, input gradient, weight and bias gradients.  fp32 mode at the
    parity gate; bf16 mode against the same convolution of the bf16-rounded operands."""
    import torch.nn.functional as TF

    import satflow_amd
    from satflow_amd import functional as F
    from satflow_amd import functional_gan as FG
    from satflow_amd._hip import cpad

    g = torch.Generator().manual_seed(cin * 3 + cout + stride)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 4, 4, generator=g) / (4 * cin**0.5)
    b = torch.randn(cout, generator=g)
    rnd = (lambda t: t.bfloat16().float()) if mode == "bf16" else (lambda t: t)
    xr, wr, br = rnd(x).clone().requires_grad_(), rnd(wt).clone().requires_grad_(), b.clone().requires_grad_()
    ref = TF.conv2d(xr, wr, br, stride=stride, padding=1)
    cot = torch.randn(ref.shape, generator=g)
    satflow_amd.set_compute_dtype(mode)
    try:
        xd = x.to(device).requires_grad_()
        wd, bd = wt.to(device).requires_grad_(), b.to(device).requires_grad_()
        xn = F.nchw_to_nhwc(xd)
        eng = F.ConvEngine([4 * cpad(cin)], cout)
        y = F.nhwc_to_nchw(FG.conv4x4_as_3x3(xn, wd, bd, stride, eng), cout)
        assert y.shape == ref.shape
        assert_close(y, ref.detach(), f"conv4x4 stride {stride} ({mode})")
        (y * cot.to(device)).sum().backward()
        if mode == "bf16":   # the kernels round the cotangent to bf16 for both gradient products; the bias gradient sums it unrounded
            dxr, dwr = torch.autograd.grad(TF.conv2d(xr, wr, None, stride=stride, padding=1), (xr, wr), rnd(cot))
            dbr = cot.sum(dim=(0, 2, 3))
        else:
            (ref * cot).sum().backward()
            dxr, dwr, dbr = xr.grad, wr.grad, br.grad
        assert_close(xd.grad, dxr, "dx", grad=True)
        assert_close(wd.grad, dwr, "dW", grad=True)
        assert_close(bd.grad, dbr, "db", grad=True)
    finally:
        satflow_amd.set_compute_dtype("f32")


@pytest.mark.parametrize("case", ["small", "rect"])
def test_cloudgan_training_steps_bf16_modes_vs_golden(device, case):
    """The two optimizer steps of the golden run in the benchmarked arithmetic (bf16 MFMA operands everywhere incl. the PatchGAN discriminator's
    4x4 convolutions, `bf16a`): losses and every parameter gradient against the reference's fp32 goldens.  Yardstick = the CPU oracle of the
    same steps under ``torch.autocast(bfloat16)`` (what the reference's `precision: 16` run computes): a parameter's gradient error may be at
    most 1.5x the yardstick's error FOR THAT PARAMETER, or 2e-2 (these B = 1 goldens are sensitive: the autocast run itself is 9-22 % off on
    some biases); the observed figures are published.  Round 6: the former escape hatch (the yardstick's worst parameter of the whole step as a
    bound for every parameter) is gone; ONE parameter of ONE golden is allow-listed with its measured cause (``ALLOW`` below)."""
    import satflow_amd
    from oracle import cloudgan as OC
    from parity_util import publish
    from satflow_amd.models import CloudGAN

    G = _load(f"cloudgan_{case}.npz")
    B, T, C, H, W = G["images"].shape
    fs, lam, nf = int(G["forecast_steps"]), float(G["lambda_l1"]), int(G["num_filters"])

    def rel(a, ref):
        return float((a.float().cpu() - ref).norm() / ref.norm())

    def yardstick(tag):
        gen = {k[len("gen."):]: v.clone().float().requires_grad_() for k, v in G.items() if k.startswith("gen.")}
        disc = {k[len("disc."):]: (v.clone().float().requires_grad_() if v.dtype == torch.float32 and "running" not in k else v.clone())
                for k, v in G.items() if k.startswith("disc.")}
        with torch.autocast("cpu", dtype=torch.bfloat16):
            if tag == "g":
                loss = OC.generator_step(G["images"], G["future"], gen, disc, fs, lam)[0]
            else:
                loss = OC.discriminator_step(G["images"], G["future"], {k: v.detach() for k, v in gen.items()}, disc, fs)[0]
        loss.float().backward()
        grads = {f"gen.{k}": v.grad for k, v in gen.items() if v.grad is not None}
        grads.update({f"disc.{k}": v.grad for k, v in disc.items() if getattr(v, "grad", None) is not None})
        return float(loss), grads

    # gen.decoder_1_convlstm.conv.bias, generator step of the 'small' golden: 18.0 % against a 9.4 % yardstick (1.9x).  tools/probe_cloudgan_bias.py
    # (profiles/r06_cloudgan_decoder1_bias.txt) ran the step under every storage switch of the stack: fp32-stored [dx ; dh] 17.9 %, fp32 head gradient
    # 18.0 %, fp32 frames 18.0 %, c' read back 18.3 %, all four 17.6 %, and the "bf16" mode - fp32 storage EVERYWHERE, only the MFMA operands rounded -
    # 17.2 %; "f32" / "f32e" 0.0000.  So it is the bf16 operand rounding of the products themselves (what any bf16-operand implementation does), drawn
    # differently from the CPU autocast run's: the same parameter on the 'rect' golden is 5.7 % against a 9.2 % yardstick (0.6x).  The gradient is a sum
    # over 6 decoder steps x 4096 pixels that cancels to a fraction of a per cent of its l1 mass.
    ALLOW = {("small", "g", "gen.decoder_1_convlstm.conv.bias"): 0.20}
    satflow_amd.set_compute_dtype("bf16a")
    try:
        m = CloudGAN(forecast_steps=fs, input_channels=C, num_filters=nf, generator_model="convlstm", norm="batch", discriminator_model="basic",
                     loss="vanilla", scheduler="cosine", lambda_l1=lam, channels_per_timestep=C, condition_time=True)
        m.generator.load_state_dict({k[len("gen."):]: v for k, v in G.items() if k.startswith("gen.")})
        m.discriminator.load_state_dict({k[len("disc."):]: v for k, v in G.items() if k.startswith("disc.")})
        m = m.to(device).train()
        batch = (G["images"].to(device), G["future"].to(device))
        rec = {"config": f"CloudGAN training steps, golden '{case}' {tuple(G['images'].shape)}", "mode": "bf16a"}
        for idx, tag in ((0, "g"), (1, "d")):
            m.zero_grad()
            out = m.training_step(batch, 0, idx)
            out["loss"].backward()
            y_loss, y_grads = yardstick(tag)
            ref_loss = float(G[f"{tag}_loss"])
            e_loss, ey_loss = abs(float(out["loss"]) - ref_loss) / abs(ref_loss), abs(y_loss - ref_loss) / abs(ref_loss)
            assert e_loss <= max(1.5 * ey_loss, 2e-2), (tag, float(out["loss"]), y_loss, ref_loss)
            worst = ("", 0.0, 0.0)
            worst_upd = ("", 0.0, 0.0)   # ... among the parameters THIS step's optimizer updates (the generator step leaves gradients on the discriminator too)
            live = [(net, pre, k, p) for net, pre in ((m.generator, "gen"), (m.discriminator, "disc")) for k, p in net.named_parameters()
                    if p.grad is not None and f"{pre}.{k}" in y_grads and float(G[f"{tag}_grad.{pre}.{k}"].float().abs().max()) >= 1e-6]
            for net, pre, k, p in live:
                    ref = G[f"{tag}_grad.{pre}.{k}"].float()
                    err, yerr = rel(p.grad, ref), rel(y_grads[f"{pre}.{k}"], ref)
                    assert err <= max(1.5 * yerr, 2e-2, ALLOW.get((case, tag, f"{pre}.{k}"), 0.0)), (tag, pre, k, err, yerr)
                    if err > worst[1]:
                        worst = (f"{pre}.{k}", err, yerr)
                    if pre == ("gen" if tag == "g" else "disc") and err > worst_upd[1]:
                        worst_upd = (f"{pre}.{k}", err, yerr)
            rec.update({f"{tag}_loss_rel": e_loss, f"{tag}_loss_rel_autocast": ey_loss, f"{tag}_step_worst_grad": worst[0],
                        f"{tag}_step_worst_grad_rel_l2": worst[1], f"{tag}_step_that_grad_autocast_rel_l2": worst[2],
                        f"{tag}_step_worst_updated_grad": worst_upd[0], f"{tag}_step_worst_updated_grad_rel_l2": worst_upd[1],
                        f"{tag}_step_that_updated_grad_autocast_rel_l2": worst_upd[2]})
        publish(rec)
    finally:
        satflow_amd.set_compute_dtype("f32")


def test_cloudgan_step_hipgraph_replay_equals_eager(device):
    """bench.py's CloudGAN workload: the generator half and the discriminator half capture into hipGraphs, and a replayed step leaves the
    parameters (and the discriminator's BatchNorm statistics) the eager step leaves from the same state."""
    import copy

    import satflow_amd
    import bench

    satflow_amd.set_compute_dtype("bf16a")
    try:
        a = bench.CloudGANWorkload(device, 2, 0)
        b = bench.CloudGANWorkload(device, 2, 0)
        a.capture()
        assert a.graphed
        b.model.load_state_dict(copy.deepcopy(a.model.state_dict()))
        for oa, ob in ((a.opt_g, b.opt_g), (a.opt_d, b.opt_d)):
            ob.load_state_dict(copy.deepcopy(oa.state_dict()))
        # THREE steps each: a captured scratch reset that only works on a graph's first launch (hipMemsetAsync nodes, round 5) shows from the second on
        for k in range(3):
            la, lb = a.step(), b._eager_step()
            torch.cuda.synchronize()
            assert_close(la, lb, f"loss of replayed step {k}", rtol=2e-3 if k else 1e-4, atol=1e-4 if k else 1e-5)
        sa, sb = a.model.state_dict(), b.model.state_dict()
        n = 0
        for k, v in sa.items():
            if v.dtype.is_floating_point:
                assert_close(v.float(), sb[k].float(), f"{k} after three replayed steps vs after three eager steps", rtol=2e-2, atol=2e-4)
                n += 1
        assert n > 20
    finally:
        satflow_amd.set_compute_dtype("f32")
