"""Time sf_convlstm_cell_fwd at the benchmark's shape (B = 8, 128 -> 256 channels, 128x128) under the library SATFLOW_HIP_LIB points to."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import satflow_amd, bench
from satflow_amd._hip import T, gate_storage_dtype, state_storage_dtype

tag = sys.argv[1] if len(sys.argv) > 1 else "base"
satflow_amd.set_compute_dtype("bf16a")
dev = torch.device("cuda:0")
wl = bench.ConvLSTMWorkload(dev, 8, 0)
eng = wl.model.model.encoder_2_convlstm.engine
B, H, W, hid = wl.B, wl.H, wl.W, wl.hid
st, gt = state_storage_dtype(), gate_storage_dtype()
mk = lambda c, dt=torch.float32: torch.randn(B, H, W, c, device=dev).to(dt)
x, h, c, ho, co, g = mk(hid, st), mk(hid, st), mk(hid), mk(hid, st), mk(hid), mk(4 * hid, gt)
flops = 2 * 9 * 2 * hid * 4 * hid * H * W * B
variants = (("training step (saved gates)", lambda: eng.step(T(x), h, c, B, H, W, ho, co, g)),
            ("no saved gates", lambda: eng.step(T(x), h, c, B, H, W, ho, co, None)))
if tag == "pmc":   # counter passes (tools/prof_pmc_cell.sh): only the training step's launch shape
    variants = variants[:1]
for name, fn in variants:
    t = bench.event_time(fn, iters=30)
    print(f"{tag:8s} {name:28s} {t * 1e6:7.1f} us  {flops / t / 1e12:7.1f} TF/s")
