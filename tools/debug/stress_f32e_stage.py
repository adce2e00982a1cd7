"""In-process hunt: idle, then a f32e MetNet forward; intermediate tensors (base, y4, feat = encoder output, GRU final state, out) compared with the first repetition."""
import os, sys, time, torch, satflow_amd
from satflow_amd.models import MetNet
satflow_amd.set_compute_dtype(sys.argv[1] if len(sys.argv) > 1 else "f32e")
dev = torch.device("cuda")
CFG3 = dict(input_channels=12, sat_channels=12, input_size=64, output_channels=12, hidden_dim=64, forecast_steps=12)
torch.manual_seed(1234)
net = MetNet(**CFG3, temporal_dropout=0.2).to(dev).train()
g = torch.Generator().manual_seed(1234)
x = torch.randn(2, 24, 12, 256, 256, generator=g).to(dev)
rec = {}
enc, rnn = net.image_encoder.module, net.temporal_enc.rnn
enc_run, rnn_run = enc.run, rnn.run
def enc_wrap(*a, **k):
    r = enc_run(*a, **k); rec["feat"] = r.detach().clone(); return r
def rnn_wrap(*a, **k):
    r = rnn_run(*a, **k); rec["gru"] = r[1][-1].detach().clone(); return r
enc.run, rnn.run = enc_wrap, rnn_wrap
ref = None
idle = float(os.environ.get("IDLE", "3"))
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    enc.capture = {}
    torch.cuda.synchronize(); time.sleep(idle)
    torch.manual_seed(4242)
    with torch.no_grad() if os.environ.get("NOGRAD") else torch.enable_grad():
        out = net(x)
    torch.cuda.synchronize()
    cur = {"base": enc.capture["base"].clone(), "y4": enc.capture["y4"].detach().clone(), "feat": rec["feat"], "gru": rec["gru"], "out": out.detach().clone()}
    if ref is None:
        ref = cur; continue
    rels = {k: float((cur[k].float() - ref[k].float()).norm() / ref[k].float().norm()) for k in cur}
    if max(rels.values()) > 2e-6:
        d = (cur["gru"] - ref["gru"]).abs().amax(dim=(1, 2, 3))
        print("repetition", it, {k: "%.2e" % v for k, v in rels.items()}, "GRU maps off:", [int(i) for i in torch.nonzero(d > 1e-5).flatten()][:8])
        f = (cur["feat"] - ref["feat"]).abs().amax(dim=(1, 2, 3)); print("   feat images off:", [int(i) for i in torch.nonzero(f > 1e-5).flatten()][:12])
        y = (cur["y4"] - ref["y4"]).abs().amax(dim=(1, 2, 3)); print("   y4 images off:", [int(i) for i in torch.nonzero(y > 1e-5).flatten()][:12])
print("done")
