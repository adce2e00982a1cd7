"""Per-step ConvGRU forward (24 steps, 24 maps of 16x16, hidden 64) on a fixed input, idle gap before every repetition; hs compared bit for bit with the first repetition."""
import os, sys, time, torch, satflow_amd
from satflow_amd.models import MetNet
mode = sys.argv[1] if len(sys.argv) > 1 else "f32e"
satflow_amd.set_compute_dtype(mode)
dev = torch.device("cuda")
CFG3 = dict(input_channels=12, sat_channels=12, input_size=64, output_channels=12, hidden_dim=64, forecast_steps=12)
torch.manual_seed(1234)
net = MetNet(**CFG3, temporal_dropout=0.2).to(dev).train()
rnn = net.temporal_enc.rnn
g = torch.Generator().manual_seed(7)
feat = torch.randn(24 * 24, 16, 16, 256, generator=g).to(dev)
idle = float(os.environ.get("IDLE", "1"))
ref, bad = None, 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
for it in range(reps):
    torch.cuda.synchronize(); time.sleep(idle)
    with torch.no_grad():
        seq, last = rnn.run(feat, 24, 24, input_dropout_done=True)
    torch.cuda.synchronize()
    hs = seq if torch.is_tensor(seq) else last[-1]
    cur = last[-1].clone()
    if ref is None:
        ref = cur; continue
    if not torch.equal(cur, ref):
        d = (cur - ref).abs().amax(dim=(1, 2, 3)); bad += 1
        print("repetition", it, "maps off:", [(int(i), "%.1e" % float(d[i])) for i in torch.nonzero(d > 0).flatten()][:6])
print("done", mode, "bad", bad, "of", reps - 1, "persistent" if not os.environ.get("SF_GRU_PER_STEP") else "per-step forced")
