"""One process = one f32 forward then one f32e forward of the full-size MetNet (dropout 0.2, capture mode, the test's seeds); the f32e output is compared with
/tmp/ref_out.pt (written by the first process).  Run many times: a process whose output is off shows the per-process flake."""
import os, sys, torch, satflow_amd
from satflow_amd.models import MetNet
dev = torch.device("cuda")
CFG3 = dict(input_channels=12, sat_channels=12, input_size=64, output_channels=12, hidden_dim=64, forecast_steps=12)
g = torch.Generator().manual_seed(1234)
x = torch.randn(2, 24, 12, 256, 256, generator=g).to(dev); cot = torch.randn(2, 12, 12, 16, 16, generator=g).to(dev)
outs = {}
for mode in ("f32", "f32e"):
    satflow_amd.set_compute_dtype(mode)
    torch.manual_seed(1234)
    net = MetNet(**CFG3, temporal_dropout=0.2).to(dev).train()
    net.image_encoder.module.capture = {}
    if os.environ.get("IDLE"):
        import time
        torch.cuda.synchronize(); time.sleep(float(os.environ["IDLE"]))   # let the GPU go idle (the test's CPU oracle runs ~10 s here)
    torch.manual_seed(4242)
    out = net(x)
    (out * cot).sum().backward()
    torch.cuda.synchronize()
    outs[mode] = out.detach().cpu()
ref_p = "/tmp/ref_out.pt"
if not os.path.exists(ref_p):
    torch.save(outs, ref_p); print("reference written")
else:
    ref = torch.load(ref_p)
    for mode in outs:
        d = outs[mode] - ref[mode]
        rel = float(d.norm() / ref[mode].norm())
        per = d.abs().amax(dim=(2, 3, 4))   # [B, L]
        flag = "  <<<<<< OFF" if rel > 1e-5 else ""
        print(mode, "rel L2 vs reference process %.3e" % rel, "max abs %.3e" % float(d.abs().max()), "worst (b, lead)", int(per.argmax()) // 12, int(per.argmax()) % 12, flag)
