"""Can the MetNet (or, SF_PROBE_WL=convlstm, the ConvLSTM) training step (forward + loss + backward) be captured into a hipGraph, and what would a replay cost?  DIAGNOSTIC: a replay repeats the
captured dropout masks (host-drawn seeds are kernel arguments).  Usage: python tools/probe_graph_step.py [fwd|fwdbwd]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import satflow_amd
import bench

what = sys.argv[1] if len(sys.argv) > 1 else "fwdbwd"
satflow_amd.set_compute_dtype("bf16a")
dev = torch.device("cuda:0")
wl = bench.ConvLSTMWorkload(dev, 8, 0) if os.environ.get("SF_PROBE_WL") == "convlstm" else bench.MetNetWorkload(dev, 8, 0)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        wl.step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    wl.step()
torch.cuda.synchronize()
print(f"eager: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step", flush=True)
g = torch.cuda.CUDAGraph()
wl.opt.zero_grad()
print("capturing", what, flush=True)
with torch.cuda.graph(g):
    loss = wl.model.training_step((wl.x, wl.y), 0)
    if what == "fwdbwd":
        loss.backward()
print("captured", flush=True)
wl.opt.step()
torch.cuda.synchronize()


def run():
    wl.opt.zero_grad()
    g.replay()
    if what == "fwdbwd":
        wl.opt.step()


for _ in range(3):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    run()
torch.cuda.synchronize()
print(f"replay ({what}): {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step, loss {float(loss):.5f}", flush=True)
