"""GPU-only determinism check of the f32e MetNet forward + backward with dropout (same seeds every repetition): outputs and gradients must repeat bit for bit."""
import sys, torch, satflow_amd
from satflow_amd.models import MetNet
mode = sys.argv[1] if len(sys.argv) > 1 else "f32e"
capture = len(sys.argv) > 2 and sys.argv[2] == "capture"
satflow_amd.set_compute_dtype(mode)
dev = torch.device("cuda")
CFG3 = dict(input_channels=12, sat_channels=12, input_size=64, output_channels=12, hidden_dim=64, forecast_steps=12)
torch.manual_seed(1234)
net = MetNet(**CFG3, temporal_dropout=0.2).to(dev).train()
g = torch.Generator().manual_seed(1234)
x = torch.randn(2, 24, 12, 256, 256, generator=g).to(dev); cot = torch.randn(2, 12, 12, 16, 16, generator=g).to(dev)
ref = None
for it in range(int(sys.argv[3]) if len(sys.argv) > 3 else 24):
    for p in net.parameters(): p.grad = None
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d): m.reset_running_stats()
    if capture: net.image_encoder.module.capture = {}
    junk = torch.randn(int(torch.randint(1, 64, (1,))), 1024, 1024, device=dev)   # vary allocator state / timing (host RNG use BEFORE the seed)
    torch.manual_seed(4242)
    out = net(x)
    (out * cot).sum().backward()
    torch.cuda.synchronize()
    cur = [out.detach().clone()] + [p.grad.detach().clone() for p in net.parameters()]
    names = ["out"] + [n for n, _ in net.named_parameters()]
    if ref is None:
        ref = cur; continue
    diffs = [(n, float((a - b).abs().max()), float((a - b).norm() / (b.norm() + 1e-30))) for n, a, b in zip(names, cur, ref) if not torch.equal(a, b)]
    if diffs:
        print("repetition", it, "differs in", len(diffs), "tensors; out:", [d for d in diffs if d[0] == "out"], "worst:", max(diffs, key=lambda d: d[2]))
print("done", mode, "capture" if capture else "")
