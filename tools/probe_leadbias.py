"""Timing probe of sf_leadtime_pool_fwd/bwd at the MetNet cfg3 shape (192 frames 64x64x160, 12 lead times)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, satflow_amd
from satflow_amd.functional import leadtime_pool
dev = torch.device("cuda:0")
st = torch.bfloat16 if os.environ.get("SF_ACT", "bf16") == "bf16" else torch.float32
base = torch.randn(192, 64, 64, 160, device=dev).to(st).requires_grad_()
w = (torch.randn(160, 108, 3, 3, device=dev) * 0.1).requires_grad_()
y = leadtime_pool(base, w, 96, 12)
g = torch.randn_like(y)
def run():
    base.grad = None; w.grad = None
    y.backward(g, retain_graph=True)
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print(f"leadtime_pool bwd ({st}): {e0.elapsed_time(e1)/10:.3f} ms (incl. table/reduce kernels and torch glue)")
def fwd():
    return leadtime_pool(base.detach(), w.detach(), 96, 12, want_stats=True)  # what the MetNet step runs (sf_leadtime_pool_fwd_stats)
for _ in range(3): fwd()
e0.record()
for _ in range(10): fwd()
e1.record(); torch.cuda.synchronize()
print(f"leadtime_pool fwd ({st}): {e0.elapsed_time(e1)/10:.3f} ms")
