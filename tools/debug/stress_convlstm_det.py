"""Bit-for-bit repeatability of the ConvLSTM encoder-decoder forward (no atomics on that path) with idle gaps, per mode."""
import sys, time, torch, satflow_amd
from satflow_amd.models import EncoderDecoderConvLSTM
dev = torch.device("cuda")
for mode in sys.argv[1:] or ["bf16a", "bf16", "f32e", "f32"]:
    satflow_amd.set_compute_dtype(mode)
    torch.manual_seed(5)
    net = EncoderDecoderConvLSTM(hidden_dim=64, input_channels=12, out_channels=12, forecast_steps=6).to(dev).train()
    x = torch.randn(8, 12, 12, 128, 128, generator=torch.Generator().manual_seed(1)).to(dev)
    ref, bad = None, 0
    for it in range(25):
        torch.cuda.synchronize(); time.sleep(0.3)
        with torch.no_grad():
            y = net(x, future_seq=6)
        torch.cuda.synchronize()
        if ref is None: ref = y.clone()
        elif not torch.equal(y, ref):
            bad += 1; print(mode, "repetition", it, "max diff", float((y.float() - ref.float()).abs().max()))
    print(mode, "bad", bad, "of 24")
