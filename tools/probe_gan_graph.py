"""CloudGAN step: eager against a hipGraph replay of the two half-steps (SF_LSTM_CAPTURE_STREAMS=1: with the generator's two-stream schedules forked / joined inside
the capture).  Round 5: eager 8.1-8.3 ms, replay 8.6 (serial) / 8.45 (streams captured) - the bench line stays eager."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, satflow_amd, bench
satflow_amd.set_compute_dtype("bf16a")
dev = torch.device("cuda:0")
for graphed in (False, True, False, True):
    wl = bench.CloudGANWorkload(dev, 8, 0)
    if graphed: wl.capture()
    for _ in range(20): wl.step()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(40): wl.step()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/40
    print("graphed" if graphed else "eager  ", "%.3f ms  %.1f samples/s" % (dt*1e3, 8/dt))
    del wl
