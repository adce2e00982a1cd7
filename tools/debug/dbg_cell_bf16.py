import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, satflow_amd
from satflow_amd._hip import T
from satflow_amd.models.layers.ConvLSTM import ConvLSTMCell
satflow_amd.set_compute_dtype("bf16a")
dev = torch.device("cuda:0")
_r = lambda t: t.bfloat16().float()
for (cin, hid, n, h, w) in [(4, 8, 2, 32, 32), (4, 8, 2, 20, 24), (12, 64, 2, 20, 24), (4, 8, 1, 16, 16), (4, 16, 2, 32, 32), (4, 32, 2, 32, 32)]:
    g = torch.Generator().manual_seed(5)
    cell = ConvLSTMCell(cin, hid, (3, 3), True).to(dev)
    eng = cell.engine
    rnd = lambda *s: _r(torch.randn(*s, generator=g)).to(dev)
    x, h0 = rnd(n, h, w, eng.cinp), rnd(n, h, w, eng.hidp)
    x[..., cin:] = 0; h0[..., hid:] = 0
    c0 = torch.randn(n, h, w, eng.hidp, generator=g).to(dev)
    out = {}
    for st in (torch.float32, torch.bfloat16):
        h1 = torch.zeros(n, h, w, eng.hidp, device=dev, dtype=st); c1 = torch.zeros(n, h, w, eng.hidp, device=dev)
        gates = torch.zeros(n, h, w, 4 * eng.hidp, device=dev, dtype=st)
        xs, hs = x.to(st), h0.to(st)  # keep the converted copies alive: a descriptor does not own its tensor
        eng.step(T(xs), hs, c0, n, h, w, h1, c1, gates)
        out[st] = (h1, c1, gates)
    # torch reference on the same (bf16-representable) operands
    W = _r(cell.conv.weight.detach()); b = cell.conv.bias.detach()
    z = torch.nn.functional.conv2d(torch.cat([x[..., :cin], h0[..., :hid]], -1).permute(0, 3, 1, 2), W, b, padding=1)
    i, f, o, gg = torch.split(z, hid, 1)
    cref = (torch.sigmoid(f) * c0[..., :hid].permute(0, 3, 1, 2) + torch.sigmoid(i) * torch.tanh(gg)).permute(0, 2, 3, 1)
    for st in out:
        d = (out[st][1][..., :hid] - cref).abs()
        bad = (d > 1e-2).nonzero()
        print((cin, hid, n, h, w), st, "max err c vs torch", float(d.max()), "bad", len(bad), bad[:3].tolist())
