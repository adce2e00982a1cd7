import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "tests")
from conftest import rel_l2
from oracle import dgmr as OD
from satflow_amd.models.layers.Generator import Generator
def run(B, frames, ch, verbose=False):
    torch.manual_seed(5 + ch)
    G = Generator(in_dim=6, latent_dim=4, n_class=3, ch=ch, n_frames=frames).train()
    gen = torch.Generator().manual_seed(9)
    with torch.no_grad():
        for k, p in G.named_parameters():
            if k.endswith("bias"): p.copy_(torch.randn(p.shape, generator=gen) * 0.2)
            elif "embed.weight" in k: p.add_(torch.randn(p.shape, generator=gen) * 0.2)
    P = {k: v.detach().clone() for k, v in G.state_dict().items()}
    noise = torch.randn(B, 6, generator=gen); cls = torch.tensor([2, 0, 1][:B]); cot = torch.randn(B, frames, 3, 64, 64, generator=gen)
    P64 = {k: (v.detach().double().requires_grad_(not k.endswith(("_u", "_v")) and "running" not in k) if v.dtype.is_floating_point else v) for k, v in P.items()}
    n64 = noise.double().requires_grad_()
    ref64 = OD.generator(n64, cls, P64, None, ch=ch, latent_dim=4, n_frames=frames)
    (ref64 * cot.double()).sum().backward()
    P32 = {k: (v.detach().clone().requires_grad_(not k.endswith(("_u", "_v")) and "running" not in k) if v.dtype.is_floating_point else v) for k, v in P.items()}
    n32 = noise.clone().requires_grad_()
    ref32 = OD.generator(n32, cls, P32, None, ch=ch, latent_dim=4, n_frames=frames)
    (ref32 * cot).sum().backward()
    y = {k: rel_l2(P32[k].grad, P64[k].grad) for k in P32 if isinstance(P32[k], torch.Tensor) and P32[k].requires_grad and P32[k].grad is not None and float(P64[k].grad.norm()) > 1e-6}
    wy = max(y, key=y.get)
    print(f"   fp32 oracle: out {rel_l2(ref32, ref64):.2e} dnoise {rel_l2(n32.grad, n64.grad):.2e} worst {wy} {y[wy]:.2e}")
    G = G.cuda()
    nd = noise.cuda().requires_grad_()
    out = G(nd, cls.cuda())
    (out * cot.cuda()).sum().backward()
    errs = {k: rel_l2(p.grad, P64[k].grad) for k, p in G.named_parameters() if p.requires_grad and P64[k].grad is not None and float(P64[k].grad.norm()) > 1e-6}
    worst = max(errs, key=errs.get)
    print(f"B={B} T={frames} ch={ch}: out {rel_l2(out, ref64):.2e} dnoise {rel_l2(nd.grad, n64.grad):.2e} worst {worst} {errs[worst]:.2e}; last-block conv1 {errs['conv.11.conv1.module.weight_bar']:.2e} colorize {errs['colorize.module.weight_bar']:.2e}")
    if verbose:
        for k, e in errs.items(): print(f"   {k:55s} {e:.2e}")
for cfg in ((1, 2, 8), (1, 3, 4), (2, 1, 4), (2, 2, 4), (2, 3, 4), (3, 2, 4)):
    run(*cfg)
