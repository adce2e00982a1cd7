import sys; sys.path.insert(0, "/root/repo")
import torch, satflow_amd, bench
satflow_amd.set_compute_dtype("bf16a")
print(bench.convgru_seq_figures(torch.device("cuda:0"), 24, 96, 64))
