"""Timing probe of the persistent ConvGRU sequence kernels at MetNet's size (GPU box only)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, satflow_amd, bench
satflow_amd.set_compute_dtype("bf16a")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 96   # maps (MetNet B = 8: 96; 128 maps = two workgroups on every one of the 256 CUs)
print(bench.convgru_seq_figures(torch.device("cuda:0"), 24, n, 64))
