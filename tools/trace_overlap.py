"""Overlap analysis of a rocprofv3 --kernel-trace CSV (tools/trace_overlap.sh): over the LAST THIRD of the trace, the time with 1 / >= 2 kernels in flight,
the kernel-name pairs that overlap most, and per queue the busy time."""
import collections, csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
rows = rows[2 * n // 3:]
short = lambda s: s.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:60]
t0 = int(rows[0]["Start_Timestamp"])
ev = []
for i, r in enumerate(rows):
    ev.append((int(r["Start_Timestamp"]), 1, i))
    ev.append((int(r["End_Timestamp"]), -1, i))
ev.sort()
live, last, by_depth, pair = set(), ev[0][0], collections.Counter(), collections.Counter()
for t, d, i in ev:
    dt = t - last
    if dt > 0:
        by_depth[min(len(live), 3)] += dt
        if len(live) >= 2:
            names = sorted(short(rows[j]["Kernel_Name"]) for j in live)
            pair[(names[0], names[1])] += dt
    last = t
    if d > 0:
        live.add(i)
    else:
        live.discard(i)
span = ev[-1][0] - ev[0][0]
print(f"window {span / 1e6:.3f} ms, {len(rows)} kernels; in flight: 0 -> {by_depth[0] / span:.3f}, 1 -> {by_depth[1] / span:.3f}, 2 -> {by_depth[2] / span:.3f}, 3+ -> {by_depth[3] / span:.3f}")
qs = collections.Counter()
for r in rows:
    qs[r.get("Queue_Id", "?")] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("kernel time per queue (ms):", {k: round(v / 1e6, 3) for k, v in qs.items()})
print("top overlapping pairs (ms):")
for (a, b), v in pair.most_common(8):
    print(f"  {v / 1e6:8.3f}  {a}  ||  {b}")
# what co-running costs: mean duration of a kernel name when >= half of its run time is shared with another kernel, against running alone
iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows]
stat = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
for i, (s0, e0, nm) in enumerate(iv):
    shared = 0
    for j in range(max(0, i - 8), min(len(iv), i + 9)):
        if j != i:
            shared += max(0, min(e0, iv[j][1]) - max(s0, iv[j][0]))
    k = 2 if shared >= 0.5 * (e0 - s0) else 0
    stat[nm][k] += 1
    stat[nm][k + 1] += e0 - s0
print("mean duration alone / co-running (us), kernels with >= 10 co-running launches:")
for nm, (na, ta, nc, tc) in sorted(stat.items(), key=lambda kv: -kv[1][3]):
    if nc >= 10 and na >= 3:
        print(f"  {ta / na / 1e3:8.1f} ({na:4d})  {tc / nc / 1e3:8.1f} ({nc:4d})  {nm}")
print("timeline slice (us since window start; queue; duration):")
mid = len(rows) // 2
for r in rows[mid:mid + 24]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"  {(s - t0) / 1e3:10.1f}  q{r.get('Queue_Id', '?'):>3}  {(e - s) / 1e3:8.1f}  {short(r['Kernel_Name'])}")
