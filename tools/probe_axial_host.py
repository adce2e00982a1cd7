"""Where do the 570 us of the axial-attention layer's forward + backward go (315 -> 398 -> 570 us over three rounds with unchanged core kernels)?
Wall time (HIP events around the autograd calls, as bench.py's `attention_mfma`) of the layer at the benchmark's shape under the candidate switches, the
device time of its kernels, and a cProfile of the host side.      python tools/probe_axial_host.py      (GPU)"""
import cProfile, io, os, pstats, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import satflow_amd
from satflow_amd.models.metnet import AxialAttention
from satflow_amd.optim import FlatAdam

dev = torch.device("cuda:0")
satflow_amd.set_compute_dtype("bf16a")
n, s, hid = 96, 16, 64
torch.manual_seed(0)
layer = AxialAttention(dim=hid, dim_index=1, heads=8, num_dimensions=2).to(dev)
x = torch.randn(n, s, s, hid, device=dev).requires_grad_()


def wall(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    host = (time.perf_counter() - t0) / iters * 1e6
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3, host


def fb():
    y = layer.run(x)
    y.backward(y.detach())


def fwd():
    with torch.no_grad():
        layer.run(x)


def report(tag):
    ev, host = wall(fb)
    evf, hostf = wall(fwd)
    print(f"{tag:70s} fwd+bwd {ev:7.1f} us (host enqueue {host:7.1f})   fwd only {evf:6.1f} us (host {hostf:6.1f})", flush=True)


report("bare module (no optimizer): param_blocks route")
os.environ["SF_NO_PARAM_BLOCKS"] = "1"
report("bare module, SF_NO_PARAM_BLOCKS=1 (torch.cat route of round 3/4)")
del os.environ["SF_NO_PARAM_BLOCKS"]
opt = FlatAdam(layer.parameters(), lr=1e-3)
report("registered with FlatAdam, no zero_grad between passes (bench.py r05)")


def fbz():
    opt.zero_grad()
    y = layer.run(x)
    y.backward(y.detach())


ev, host = wall(fbz)
print(f"{'registered with FlatAdam, zero_grad before every pass (sink open)':70s} fwd+bwd {ev:7.1f} us (host enqueue {host:7.1f})", flush=True)

# device time of the layer's kernels (profiler), and the host profile
from torch.profiler import ProfilerActivity, profile

with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(10):
        fbz()
    torch.cuda.synchronize()
ka = [e for e in prof.key_averages() if e.device_time_total > 0]
tot = sum(e.device_time_total for e in ka) / 10
print(f"device time of the layer's kernels per fwd+bwd: {tot:.1f} us over {sum(e.count for e in ka) / 10:.0f} launches")
for e in sorted(ka, key=lambda e: -e.device_time_total)[:14]:
    print(f"   {e.device_time_total / 10:7.1f} us  x{e.count / 10:4.1f}  {e.key[:110]}")
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    fbz()
pr.disable()
torch.cuda.synchronize()
sio = io.StringIO()
pstats.Stats(pr, stream=sio).sort_stats("tottime").print_stats(18)
print("\n".join(l for l in sio.getvalue().splitlines() if l.strip())[:6000])
