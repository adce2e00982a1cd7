"""One shape (256->256 @32x32, 2304 images, bf16 storage) of sf_conv3x3_fwd for PMC passes."""
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
w = torch.randn(cout, cin, 3, 3, device=dev) * 0.02; b = torch.randn(cout, device=dev)
packed, bp = K.pack_weights(w, b, eng.fwd_map, False)
x = torch.randn(n, H, W, cin, device=dev).to(st); y = torch.empty(n, H, W, cout, device=dev, dtype=st)
for _ in range(4):
    K.conv3x3(T(x), NULL, n, H, W, packed, bp, eng.fwd_map, T(y))
torch.cuda.synchronize()
print("done")
