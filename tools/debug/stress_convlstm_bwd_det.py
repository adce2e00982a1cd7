"""Bit-for-bit repeatability of the ConvLSTM encoder-decoder training step's gradients with idle gaps, per mode (report only)."""
import sys, time, torch, satflow_amd
from satflow_amd.models import EncoderDecoderConvLSTM
dev = torch.device("cuda")
for mode in sys.argv[1:] or ["bf16a", "bf16", "f32e", "f32"]:
    satflow_amd.set_compute_dtype(mode)
    torch.manual_seed(5)
    net = EncoderDecoderConvLSTM(hidden_dim=64, input_channels=12, out_channels=12, forecast_steps=6).to(dev).train()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(8, 12, 12, 128, 128, generator=g).to(dev); cot = torch.randn(8, 12, 6, 128, 128, generator=g).to(dev)
    ref, bad, worst = None, 0, ("", 0.0)
    for it in range(12):
        for p in net.parameters(): p.grad = None
        torch.cuda.synchronize(); time.sleep(0.3)
        y = net(x, future_seq=6)
        (y * cot).sum().backward()
        torch.cuda.synchronize()
        cur = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
        if ref is None: ref = cur; continue
        d = [(n, float((cur[n] - ref[n]).norm() / (ref[n].norm() + 1e-30))) for n in cur if not torch.equal(cur[n], ref[n])]
        if d:
            bad += 1; worst = max([worst] + d, key=lambda t: t[1])
    print(mode, "repetitions with a differing gradient:", bad, "of 11; worst", worst)
