"""Timing probe: the axial-attention core at MetNet's size (96 maps of 16x16, hidden 64, 8 heads); SF_ATTN_NO_MFMA=1: the thread-per-row kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, satflow_amd, bench
from satflow_amd import kernels as K
dev = torch.device("cuda:0")
n, h, w, hid = 96, 16, 16, 64
qkv = torch.randn(n, h, w, 6 * hid, device=dev)
datt = torch.randn(n, h, w, 2 * hid, device=dev)
att = K.attention_core_fwd(qkv, hid, 8)
tf = bench.event_time(lambda: K.attention_core_fwd(qkv, hid, 8), iters=20)
tb = bench.event_time(lambda: K.attention_core_bwd(qkv, datt, hid, 8), iters=20)
print(f"attention core fwd {tf*1e6:.1f} us  bwd {tb*1e6:.1f} us ({'thread-per-row' if os.environ.get('SF_ATTN_NO_MFMA') else 'MFMA'})")
