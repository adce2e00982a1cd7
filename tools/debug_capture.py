"""Run the DGMR workload's hipGraph capture and print the traceback of whatever breaks it."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import satflow_amd, bench
satflow_amd.set_compute_dtype("bf16")
wl = bench.DGMRWorkload(torch.device("cuda:0"), 2, 0)
try:
    wl.capture()
    print("capture ok")
except Exception:
    traceback.print_exc()
