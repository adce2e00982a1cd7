"""The DGMR line's roofline launch (spatial discriminator stem: 128 -> 128 channels, 3x3, 16 frames of 256x256, fp32-stored activations) on its own:
the program the PMC passes of tools/prof_pmc_dgmr.sh profile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import satflow_amd, bench
satflow_amd.set_compute_dtype("bf16")
wl = bench.DGMRWorkload.__new__(bench.DGMRWorkload)   # only the fields roofline() reads: no networks built
wl.B, wl.T, wl.H, wl.chn, wl.dev = 2, 8, 256, 64, torch.device("cuda:0")
class _Stub:  # roofline() only looks the layer up
    pre_conv = [None, None, None]
wl.Ds = _Stub()
print({k: v for k, v in wl.roofline().items() if k in ("achieved", "frac", "us_per_launch")})
