"""Bit-for-bit repeatability of the MetNet forward with idle gaps: eval mode (running statistics: no atomics on the path) and train mode under no_grad, per mode."""
import sys, time, torch, satflow_amd
from satflow_amd.models import MetNet
dev = torch.device("cuda")
CFG3 = dict(input_channels=12, sat_channels=12, input_size=64, output_channels=12, hidden_dim=64, forecast_steps=12)
x = torch.randn(8, 24, 12, 256, 256, generator=torch.Generator().manual_seed(1)).to(dev)
for mode in sys.argv[1:] or ["bf16a", "bf16", "f32e"]:
    satflow_amd.set_compute_dtype(mode)
    torch.manual_seed(5)
    net = MetNet(**CFG3).to(dev)
    for train in (False, True):
        net.train(train)
        ref, bad, worst = None, 0, 0.0
        for it in range(16):
            torch.cuda.synchronize(); time.sleep(0.4)
            torch.manual_seed(11)
            with torch.no_grad():
                y = net(x)
            torch.cuda.synchronize()
            if ref is None: ref = y.clone()
            elif not torch.equal(y, ref):
                bad += 1; worst = max(worst, float((y - ref).norm() / ref.norm()))
        print(mode, "train" if train else "eval", "bad", bad, "of 15", "worst rel L2 %.2e" % worst)
