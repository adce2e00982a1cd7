"""Runs ONLY bench.py's roofline probe (the dominant kernel, same shapes) a few times - used under rocprofv3 --pmc.
argv[1]: compute mode (bf16a | bf16 | f32), default bf16a."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import satflow_amd
import bench

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16a"
satflow_amd.set_compute_dtype(mode)
dev = torch.device("cuda:0")
w = bench.MetNetWorkload.__new__(bench.MetNetWorkload)
w.B, w.T, w.L, w.dev = 8, 24, 12, dev
r = w.roofline()
print({k: r[k] for k in ("achieved", "launch_us", "algorithmic_bytes")})
