import os, sys
sys.path.insert(0, os.getcwd())
import torch, satflow_amd
from satflow_amd import kernels as K
from satflow_amd._hip import T, cpad
from satflow_amd.functional import ConvEngine
dev = torch.device("cuda:0"); satflow_amd.set_compute_dtype("bf16a")
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
cin, cout, n, H, W, G = 160, 256, 2304, 32, 32, 12
eng = ConvEngine([cin], cout); gmb = eng.bwd_map((True,))
w = torch.randn(cout, cin, 3, 3, device=dev) * 0.03
packed_t = eng.packed(w, None, "bwd", (True,))[0]
gy = torch.randn(n, H, W, eng.coutp, device=dev).to(torch.bfloat16)
x = torch.randn(n, H, W, cpad(cin), device=dev).to(torch.bfloat16)
coef = torch.randn(G, 3, cpad(cin), device=dev); dx = torch.empty_like(x)
ms = timeit(lambda: K.conv3x3_bwd_data_bn(T(gy), n, H, W, packed_t, gmb, T(x), coef, T(dx)))
print(f"{os.environ.get('SATFLOW_HIP_LIB','default')[-20:]} NF5 dgrad+BNB 256->160: {ms:.3f} ms nf={gmb.nf}")
