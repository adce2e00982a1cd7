"""Per-kernel HBM traffic of a tools/prof_pmc_step.sh run: python tools/pmc_by_kernel.py gpurun_out/<name> [top]"""
import collections, csv, sys
d, top = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = collections.defaultdict(lambda: [0, 0.0, 0.0])
for c, scale, col in (("FETCH_SIZE", 2.0, 1), ("WRITE_SIZE", 1.0, 2)):   # FETCH_SIZE counts 32-byte... corrected x2 on gfx950 (MI355X_MICROARCH.md)
    for r in csv.DictReader(open(f"{d}/pmc_{c}_counter_collection.csv")):
        k = r["Kernel_Name"][:110]
        tot[k][col] += float(r["Counter_Value"]) * 1024 * scale
        if col == 1:
            tot[k][0] += 1
print(f"{'launches':>8} {'read GB/launch':>15} {'write GB/launch':>16}  kernel")
for k, (n, rd, wr) in sorted(tot.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:top]:
    n = max(n, 1)
    print(f"{n:8d} {rd / n / 1e9:15.3f} {wr / n / 1e9:16.3f}  {k}")
