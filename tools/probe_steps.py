"""Per-step wall times of the first MetNet steps (synchronised after every step): where do outliers come from?  argv[1] = "nogc": with the cyclic
garbage collector disabled after the first two steps."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, satflow_amd, bench
satflow_amd.set_compute_dtype("bf16a")
dev = torch.device("cuda:0")
wl = bench.MetNetWorkload(dev, 8, 0)
ts = []
for i in range(60):
    if i == 2 and len(sys.argv) > 1 and sys.argv[1] == "nogc":
        gc.collect(); gc.disable()
    torch.cuda.synchronize(); t0 = time.perf_counter(); wl.step(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("per-step ms:", [round(t, 1) for t in ts[1:]])
print("outliers (> 26 ms):", [(i + 1, round(t, 1)) for i, t in enumerate(ts[1:]) if t > 26], "gc counts", gc.get_count())
