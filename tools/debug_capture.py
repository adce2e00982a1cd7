"""Run a workload's hipGraph capture and print the traceback of whatever breaks it:  python tools/debug_capture.py dgmr|cloudgan"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import satflow_amd, bench
name = sys.argv[1] if len(sys.argv) > 1 else "dgmr"
satflow_amd.set_compute_dtype("bf16a" if name == "cloudgan" else "bf16")
wl = bench.CloudGANWorkload(torch.device("cuda:0"), 8, 0) if name == "cloudgan" else bench.DGMRWorkload(torch.device("cuda:0"), 2, 0)
try:
    wl.capture()
    print("capture ok")
except Exception:
    traceback.print_exc()
