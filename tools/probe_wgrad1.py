"""One shape (256->256 @32x32, 2304 images) of sf_conv3x3_bwd_weight for PMC passes.  SF_ACT=bf16|f32."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, satflow_amd
from satflow_amd import kernels as K
from satflow_amd._hip import T, NULL
from satflow_amd.functional import ConvEngine
satflow_amd.set_compute_dtype("bf16")
dev = torch.device("cuda:0")
st = torch.bfloat16 if os.environ.get("SF_ACT", "bf16") == "bf16" else torch.float32
n, cin, cout, H, W = 2304, 256, 256, 32, 32
eng = ConvEngine([cin], cout)
x = torch.randn(n, H, W, cin, device=dev).to(st); gy = torch.randn(n, H, W, cout, device=dev).to(st)
dw = torch.empty(cout, cin, 3, 3, device=dev); db = torch.empty(cout, device=dev)
for _ in range(4):
    K.conv3x3_bwd_weight(T(x), NULL, T(gy), n, H, W, eng.wgrad_map, dw, db, False)
torch.cuda.synchronize()
print("done")
