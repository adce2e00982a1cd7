import os, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch, satflow_amd
from satflow_amd import kernels as K
satflow_amd.set_compute_dtype("bf16a")
dev = torch.device("cuda:0")
x = torch.randn(8, 24, 12, 256, 256, device=dev)
f = lambda: K.metnet_preprocess(x, 12, 64, torch.bfloat16)
y = f()
for _ in range(3): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
print(os.environ.get("SATFLOW_HIP_LIB", "default")[-24:], "preprocess us", e0.elapsed_time(e1) / 20 * 1e3, float(y.float().abs().sum()))
