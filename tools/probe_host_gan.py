"""Host enqueue time vs total time of a CloudGAN / DGMR step (is the step host-bound?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import satflow_amd, bench
satflow_amd.set_compute_dtype("bf16a")
dev = torch.device("cuda:0")
for name, mk in (("cloudgan", lambda: bench.CloudGANWorkload(dev, 8, 0)),):
    wl = mk()
    for _ in range(3): wl.step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); wl.step(); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t0))
    print(name, "enqueue ms", [round(a * 1e3, 1) for a, _ in ts], "total ms", [round(b * 1e3, 1) for _, b in ts])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        wl.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(name, "10 steps back to back: enqueue", round((t1 - t0) * 100, 2), "ms/step, total", round((t2 - t0) * 100, 2), "ms/step")
    loss = None
    t0 = time.perf_counter()
    for _ in range(10):
        loss = wl.step()
    torch.cuda.synchronize()
    print(name, "10 more, keeping the loss:", round((time.perf_counter() - t0) * 100, 2), "ms/step")
