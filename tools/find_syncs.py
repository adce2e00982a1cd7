"""Does one training step of a workload synchronise with the device (tensor.item(), float(tensor), D2H copies)?  python tools/find_syncs.py <workload>
Prints the aten::_local_scalar_dense / aten::item / copy-to-CPU events of ONE step with the innermost frames of this repository."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
import satflow_amd, bench

name = sys.argv[1] if len(sys.argv) > 1 else "metnet"
satflow_amd.set_compute_dtype("bf16" if name == "dgmr" else "bf16a")
dev = torch.device("cuda:0")
os.environ["SF_NO_GRAPH"] = "1"
wl = bench.build_workload(name, dev, 2 if name == "dgmr" else 8, 0)
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    wl.step()
    torch.cuda.synchronize()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = 0
for ev in prof.events():
    if ev.name in ("aten::_local_scalar_dense", "aten::item") or (ev.name == "aten::_to_copy" and "cpu" in str(ev.input_shapes)):
        frames = [f for f in (ev.stack or []) if root in f or "satflow_amd" in f or "bench.py" in f]
        print(ev.name, " <- ".join(f.replace(root + "/", "") for f in frames[:3]))
        n += 1
print(f"{name}: {n} synchronising events in one step")
