"""Launch the dominant kernel of bench.py's roofline object a few times (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import satflow_amd
from bench import MetNetWorkload, ConvLSTMWorkload
satflow_amd.set_compute_dtype(os.environ.get("SF_DTYPE", "bf16"))
dev = torch.device("cuda:0")
wl = MetNetWorkload.__new__(MetNetWorkload)
wl.B, wl.T, wl.L, wl.dev = 8, 24, 12, dev
print(wl.roofline())
