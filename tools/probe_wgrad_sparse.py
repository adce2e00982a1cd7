"""Probe: the weight gradient of conv4 (256 -> 256 @32x32 x 2304, 12 BatchNorm groups) on a gradient that comes out of a 2x2 max-pooling - the dense
all-bf16 kernel against the 2:4 structured-sparse path (sf_conv3x3_bwd_weight_folded_sparse24): results against each other and float64, timings."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, satflow_amd
from satflow_amd import kernels as K
from satflow_amd._hip import T, cpad
from satflow_amd.functional import ConvEngine
satflow_amd.set_compute_dtype("bf16a")
dev = torch.device("cuda:0")
n, H, W, cin, cout, G = int(os.environ.get("SF_PROBE_N", "2304")), 32, 32, 256, 256, 12
torch.manual_seed(1)
eng = ConvEngine([cin], cout)
x = torch.randn(n, H, W, cin, device=dev).bfloat16()
y = torch.randn(n, H, W, cout, device=dev).bfloat16()
pooled, route = K.maxpool2_route_fwd(y, None, torch.bfloat16, None)
g = torch.randn_like(pooled)
dout = K.maxpool2_route_bwd(route, g, tuple(y.shape), torch.bfloat16, None, None)
scale = 0.5 + torch.rand(G, cpad(cin), device=dev); shift = torch.randn(G, cpad(cin), device=dev)
w = torch.randn(cout, cin, 3, 3, device=dev) * 0.03
mean, rstd = torch.randn(G, cpad(cin), device=dev), 0.5 + torch.rand(G, cpad(cin), device=dev)
def run(sparse, pooled=None):
    dw, db = torch.empty_like(w), torch.empty(cout, device=dev)
    sums = torch.empty(G, 2, cpad(cin), dtype=torch.float64, device=dev)
    f = lambda: K.conv3x3_bwd_weight_folded(T(x), T(dout), n, H, W, eng.wgrad_map, scale, shift, dw, db, bn=(w, mean, rstd, sums), pooled_gradient=sparse,
                                            pooled=pooled)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    return dw, db, sums, e0.elapsed_time(e1) / 10
d0 = run(False); d1 = run(True); d2 = run(True, (g, route, None))
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
print(f"dense {d0[3]:.3f} ms   sparse {d1[3]:.3f} ms   dW rel {rel(d1[0], d0[0]):.2e}  db rel {rel(d1[1], d0[1]):.2e}  bn sums rel {rel(d1[2], d0[2]):.2e}")
print(f"pooled operand {d2[3]:.3f} ms   dW == sparse-from-dout: {torch.equal(d2[0], d1[0])} (rel {rel(d2[0], d1[0]):.2e})  db rel {rel(d2[1], d1[1]):.2e}  bn sums equal {torch.equal(d2[2], d1[2])}")
