import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from oracle import metnet as M
from test_metnet_gpu import _metnet_pair, _g
dev = torch.device("cuda:0")
cfg = dict(input_channels=13, sat_channels=12, input_size=16, output_channels=3, hidden_dim=32, forecast_steps=4, num_att_layers=2)
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 2
net, P = _metnet_pair(dev, cfg)
raw = 64
x = torch.randn(B, T, 13, raw, raw, generator=_g(21)); cot = torch.randn(B, 4, 3, 4, 4, generator=_g(22))
kw = dict(sat_channels=12, input_size=16, forecast_steps=4, num_att_layers=2)
ref = M.metnet_forward(x, P, **kw); (ref * cot).sum().backward()
P64 = {k: v.detach().double().requires_grad_() for k, v in P.items()}
ref64 = M.metnet_forward(x.double(), P64, **kw); (ref64 * cot.double()).sum().backward()
net.train(); out = net(x.to(dev)); (out * cot.to(dev)).sum().backward()
rl = lambda a, b: float((a.double() - b).norm() / (b.norm() + 1e-300))
print(f"out: hip-vs-64 {rl(out.cpu(), ref64):.2e}  cpu32-vs-64 {rl(ref, ref64):.2e}")
for k, p in net.named_parameters():
    print(f"{k:55s} hip-vs-64 {rl(p.grad.cpu(), P64[k].grad):.2e}   cpu32-vs-64 {rl(P[k].grad, P64[k].grad):.2e}   hip-vs-cpu32 {rl(p.grad.cpu(), P[k].grad.double()):.2e}")

g = net.image_encoder.module.module[0].weight.grad.cpu(); r = P64["image_encoder.module.module.0.weight"].grad
err = ((g.double() - r).abs()).amax(dim=(0, 2, 3)); mag = r.abs().amax(dim=(0, 2, 3))
print("per-input-channel max err / max |ref|:")
print(" ".join(f"{i}:{float(e):.1e}/{float(m):.1e}" for i, (e, m) in enumerate(zip(err, mag))))
