"""Per-kernel HBM traffic of a tools/prof_pmc_step.sh run: python tools/pmc_by_kernel.py gpurun_out/<name> [top] [--json out.json]
(--json: a sha-stamped record {kernel: {launches, read_bytes, write_bytes}} per launch, what bench.py's kernel table reads)"""
import collections, csv, json, os, sys
js = None
if "--json" in sys.argv:
    i = sys.argv.index("--json"); js = sys.argv[i + 1]; del sys.argv[i:i + 2]
d, top = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = collections.defaultdict(lambda: [0, 0.0, 0.0])
seq = collections.defaultdict(lambda: ([], []))   # per kernel name: the launches' bytes in dispatch order, (reads, writes) - the two passes replay the same program
for c, scale, col in (("FETCH_SIZE", 2.0, 1), ("WRITE_SIZE", 1.0, 2)):   # FETCH_SIZE counts 32-byte... corrected x2 on gfx950 (MI355X_MICROARCH.md)
    rows = list(csv.DictReader(open(f"{d}/pmc_{c}_counter_collection.csv")))
    if rows and "Dispatch_Id" in rows[0]:
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        k = r["Kernel_Name"][:110]
        v = float(r["Counter_Value"]) * 1024 * scale
        tot[k][col] += v
        seq[k][col - 1].append(v)
        if col == 1:
            tot[k][0] += 1


def shape_classes(reads, writes):
    """A kernel NAME that runs at several shapes in a step (256->256 and 160->256 launches of the same instantiation) reports a mix when averaged by name.
    The i-th launch of a name is the same launch in both passes; launches are grouped by their bytes: writes within 6 % (nearly deterministic per shape: the statistics launches of one shape were seen 3.4 % apart) AND
    reads within 12 % (cache effects move them run to run; the shapes of one name differ by >= 25 %).  Classes are returned sorted by total bytes."""
    if len(reads) != len(writes) or not writes:
        return None
    cls = []
    for rd, wr in zip(reads, writes):
        for c in cls:
            mw, mr = c["write_bytes"] / c["launches"], c["read_bytes"] / c["launches"]
            if abs(wr - mw) <= 0.06 * max(wr, mw, 1.0) and abs(rd - mr) <= 0.12 * max(rd, mr, 1.0):
                c["launches"] += 1; c["read_bytes"] += rd; c["write_bytes"] += wr
                break
        else:
            cls.append({"launches": 1, "read_bytes": rd, "write_bytes": wr})
    out = [{"launches": c["launches"], "read_bytes": c["read_bytes"] / c["launches"], "write_bytes": c["write_bytes"] / c["launches"]} for c in cls]
    out = [c for c in out if c["launches"] >= max(2, len(writes) // 20)] or out   # (stray launches - warm-up shapes - do not make a class)
    return sorted(out, key=lambda c: c["read_bytes"] + c["write_bytes"])


print(f"{'launches':>8} {'read GB/launch':>15} {'write GB/launch':>16}  kernel")
for k, (n, rd, wr) in sorted(tot.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:top]:
    n = max(n, 1)
    print(f"{n:8d} {rd / n / 1e9:15.3f} {wr / n / 1e9:16.3f}  {k}")

if js:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_sha
    rec = {"what": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of a short bench.py run, per-launch means by kernel; FETCH_SIZE x 2 (MI355X_MICROARCH.md)",
           "source": d, "kernel_src_sha": kernel_source_sha(),
           "kernels": {k: {"launches": n, "read_bytes": rd / max(n, 1), "write_bytes": wr / max(n, 1), "shapes": shape_classes(*seq[k])}
                       for k, (n, rd, wr) in tot.items() if rd + wr > 50e6 * max(n, 1)}}
    json.dump(rec, open(js, "w"), indent=1)
