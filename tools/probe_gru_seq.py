"""Timing probe of the persistent ConvGRU sequence kernels at MetNet's size (GPU box only)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, satflow_amd, bench
satflow_amd.set_compute_dtype("bf16a")
print(bench.convgru_seq_figures(torch.device("cuda:0"), 24, 96, 64))
