"""Timing: persistent ConvGRU sequence kernel vs the per-step launches at MetNet's recurrent shape (T=24, 96 maps of 16x16, hid 64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, satflow_amd
from satflow_amd import kernels as K
from satflow_amd._hip import T
from satflow_amd.functional import GRUEngine
satflow_amd.set_compute_dtype("bf16a")
dev = torch.device("cuda:0")
Tn, n, H, W, hid = 24, int(os.environ.get("SF_N", 96)), 16, 16, 64
eng = GRUEngine(256, hid)
Wh = torch.randn(3 * hid, hid, 3, 3, device=dev) * 0.05; bh = torch.randn(3 * hid, device=dev) * 0.1
packed, bp = K.pack_weights(Wh, bh, eng.h_fwd, False)
def ev(f, it=10):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for gdt in (torch.bfloat16, torch.float32):
    gx = torch.randn(Tn * n, H, W, 3 * hid, device=dev).to(gdt)
    hs = torch.empty(Tn, n, H, W, hid, device=dev); gates = torch.empty(Tn, n, H, W, 4 * hid, device=dev, dtype=torch.bfloat16)
    t_seq = ev(lambda: K.convgru_seq_fwd(gx, None, Tn, n, H, W, packed, bp, hid, hs, gates))
    t_seq_ng = ev(lambda: K.convgru_seq_fwd(gx, None, Tn, n, H, W, packed, bp, hid, hs, None))
    print(f"persistent, gx {gdt}: {t_seq:.1f} us ({t_seq/Tn:.2f} us/step); without gates {t_seq_ng:.1f} us")
gxf = torch.randn(Tn, n, H, W, 3 * hid, device=dev)
def per_step():
    for t in range(Tn):
        K.convgru_step_fwd(T(gxf[t]), hs[t - 1] if t else None, n, H, W, packed, bp, hid, hs[t], gates[t])
t_ps = ev(per_step)
print(f"per-step launches: {t_ps:.1f} us ({t_ps/Tn:.2f} us/step)")
