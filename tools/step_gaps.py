"""Idle time INSIDE one training step from a rocprofv3 --kernel-trace CSV: the span between two consecutive launches of a once-per-step kernel (default: the
preprocess kernel of the MetNet step), the union of the kernel intervals in it, and the gaps by size class and by the kernel that follows them.
    python tools/step_gaps.py <dir>/tr_kernel_trace.csv [marker substring]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "preprocess_kernel"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
assert len(idx) >= 3, "need at least three steps in the trace"
a, b = idx[-3], idx[-2]   # a full step well inside the timed region (the last one is followed by the bench's probes)
step = rows[a:b]
t0, t1 = int(step[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
busy, cur_e, gaps = 0, t0, []
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > cur_e:
        gaps.append((s - cur_e, r["Kernel_Name"][:60]))
        cur_e = s
    if e > cur_e:
        busy += e - max(s, cur_e) if s < cur_e else e - s
        cur_e = e
if t1 > cur_e:
    gaps.append((t1 - cur_e, "(next step's first kernel)"))
tot = sum(g for g, _ in gaps)
print(f"step span {(t1 - t0) / 1e6:.3f} ms, {len(step)} kernels, idle {tot / 1e6:.3f} ms = {100 * tot / (t1 - t0):.1f} %")
for lo, hi in ((0, 2000), (2000, 5000), (5000, 10000), (10000, 20000), (20000, 10**12)):
    sel = [g for g, _ in gaps if lo <= g < hi]
    print(f"   gaps {lo / 1e3:5.0f} .. {hi / 1e3 if hi < 10**11 else float('inf'):5.0f} us: {len(sel):4d}  total {sum(sel) / 1e3:8.1f} us")
by = collections.Counter()
for g, n in gaps:
    by[n] += g
for n, g in by.most_common(12):
    print(f"   {g / 1e3:8.1f} us before  {n}")
