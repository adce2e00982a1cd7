import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from oracle import metnet as M
from test_metnet_gpu import _metnet_pair, _g
dev = torch.device("cuda:0")
cfg = dict(input_channels=5, sat_channels=4, input_size=8, output_channels=2, hidden_dim=16, forecast_steps=3)
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 2
net, P = _metnet_pair(dev, cfg)
raw = 32
x = torch.randn(B, T, 5, raw, raw, generator=_g(21)); cot = torch.randn(B, 3, 2, 2, 2, generator=_g(22))
kw = dict(sat_channels=4, input_size=8, forecast_steps=3)
ref = M.metnet_forward(x, P, **kw); (ref * cot).sum().backward()
P64 = {k: v.detach().double().requires_grad_() for k, v in P.items()}
ref64 = M.metnet_forward(x.double(), P64, **kw); (ref64 * cot.double()).sum().backward()
net.train(); out = net(x.to(dev)); (out * cot.to(dev)).sum().backward()
rl = lambda a, b: float((a.double() - b).norm() / (b.norm() + 1e-300))
print(f"out: hip-vs-64 {rl(out.cpu(), ref64):.2e}  cpu32-vs-64 {rl(ref, ref64):.2e}")
for k, p in net.named_parameters():
    print(f"{k:55s} hip-vs-64 {rl(p.grad.cpu(), P64[k].grad):.2e}   cpu32-vs-64 {rl(P[k].grad, P64[k].grad):.2e}   hip-vs-cpu32 {rl(p.grad.cpu(), P[k].grad.double()):.2e}")
