import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch.nn.functional as TF
from oracle import metnet as M
from satflow_amd import functional as F
from test_metnet_gpu import _metnet_pair, _g
dev = torch.device("cuda:0")
cfg = dict(input_channels=13, sat_channels=12, input_size=16, output_channels=3, hidden_dim=32, forecast_steps=4, num_att_layers=2)
B, T, L = 2, 3, 4
net, P = _metnet_pair(dev, cfg)
x = torch.randn(B, T, 13, 64, 64, generator=_g(21)); cot = torch.randn(B, 4, 3, 4, 4, generator=_g(22))
# ---- oracle with captured stage outputs (per lead time)
cap_ref = []
orig_ds = M.downsampler
def ds(xx, p, prefix, bn_stats):
    conv = lambda i, t: TF.conv2d(t, p[f"{prefix}.{i}.weight"], p[f"{prefix}.{i}.bias"], padding=1)
    bn = lambda i, t: M.batch_norm_train(t, p[f"{prefix}.{i}.weight"], p[f"{prefix}.{i}.bias"])
    st = []
    def k(t): t.retain_grad(); st.append(t); return t
    t = k(conv(0, xx)); t = k(TF.max_pool2d(t, 2)); t = k(bn(3, t)); t = k(conv(4, t)); t = k(bn(5, t)); t = k(conv(6, t)); t = k(bn(7, t)); t = k(conv(8, t)); t = k(TF.max_pool2d(t, 2))
    cap_ref.append(st)
    return t
M.downsampler = ds
ref = M.metnet_forward(x, P, sat_channels=12, input_size=16, forecast_steps=4, num_att_layers=2); (ref * cot).sum().backward()
# ---- HIP with captured stage outputs
cap = []
def wrap(fn):
    def w(*a, **k):
        y = fn(*a, **k); y.retain_grad(); cap.append(y); return y
    return w
F.conv3x3_broadcast = wrap(F.conv3x3_broadcast); F.conv3x3 = wrap(F.conv3x3); F.maxpool2 = wrap(F.maxpool2); F.batchnorm = wrap(F.batchnorm)
net.train(); out = net(x.to(dev)); (out * cot.to(dev)).sum().backward()
names = ["conv1", "pool1", "bn1", "conv2", "bn2", "conv3", "bn3", "conv4", "pool4"]
order = [0, 1, 2, 3, 4, 5, 6, 7, 8]  # hip capture order: conv1(bcast), maxpool, bn, conv, bn, conv, bn, conv, maxpool(perm)
rl = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))
for si, name in enumerate(names):
    t = cap[si]
    n, h, w, c = t.shape
    for l in range(L):
        r = cap_ref[l][si]  # [T*B frames (b-major in oracle: index b*T + t), C, h, w]
        creal = r.shape[1]
        rr = r.view(B, T, creal, h, w).permute(1, 0, 3, 4, 2)  # [T,B,h,w,c]
        rg = r.grad.view(B, T, creal, h, w).permute(1, 0, 3, 4, 2)
        if name == "pool4":
            mine = t.view(T, L, B, h, w, c)[:, l]; mg = t.grad.view(T, L, B, h, w, c)[:, l]
        else:
            mine = t.view(L, T, B, h, w, c)[l]; mg = t.grad.view(L, T, B, h, w, c)[l]
        print(f"{name:6s} lead {l}: fwd {rl(mine[..., :creal].cpu(), rr):.2e}  grad {rl(mg[..., :creal].cpu(), rg):.2e}")
# ---- element-level look at conv4's output gradient, lead 0
t = cap[7]; n, h, w, c = t.shape
r = cap_ref[0][7]
rg = r.grad.view(B, T, c, h, w).permute(1, 0, 3, 4, 2)
mg = t.grad.view(L, T, B, h, w, c)[0].cpu()
d = (mg - rg).abs()
big = d > 1e-4 * rg.abs().max()
print("conv4 grad lead0: elements off:", int(big.sum()), "of", big.numel(), " nonzero ref:", int((rg != 0).sum()), " nonzero mine:", int((mg != 0).sum()))
idx = big.nonzero()[:12]
for i in idx:
    i = tuple(int(v) for v in i)
    print(i, float(mg[i]), float(rg[i]))
# forward values of conv4 output at those windows
mine_f = t.view(L, T, B, h, w, c)[0].detach().cpu(); ref_f = r.view(B, T, c, h, w).permute(1, 0, 3, 4, 2).detach()
for i in idx[:4]:
    tt, bb, yy, xx, cc = (int(v) for v in i)
    y0, x0 = yy // 2 * 2, xx // 2 * 2
    print("window mine", mine_f[tt, bb, y0:y0+2, x0:x0+2, cc].flatten().tolist(), "ref", ref_f[tt, bb, y0:y0+2, x0:x0+2, cc].flatten().tolist())
