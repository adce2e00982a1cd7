"""Summarise the rocprofv3 --pmc passes of tools/prof_pmc.sh into profiles/<name>.json (per-launch averages).

Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE/WRITE_SIZE are in KB; on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read (16 B/lane global_load and
buffer_load...lds alike) -> doubled; WRITE_SIZE is taken as is (it matches the algorithmic write exactly here)."""
import csv, json, sys
src, out, kernel = sys.argv[1], sys.argv[2], sys.argv[3]
res = {"kernel_filter": kernel, "source": src}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open(f"{src}/pmc_{c}_counter_collection.csv")))
    vals = [float(r["Counter_Value"]) for r in rows if kernel in r["Kernel_Name"] and r["Counter_Name"] == c]
    res[c + "_KB_avg"] = sum(vals) / len(vals)
    res[c + "_launches"] = len(vals)
    res["kernel_name"] = next(r["Kernel_Name"] for r in rows if kernel in r["Kernel_Name"])
res["read_bytes_corrected"] = res["FETCH_SIZE_KB_avg"] * 1024 * 2
res["write_bytes"] = res["WRITE_SIZE_KB_avg"] * 1024
res["traffic_bytes"] = res["read_bytes_corrected"] + res["write_bytes"]
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_sha
res["kernel_src_sha"] = kernel_source_sha()  # bench.py drops `traffic` when the kernel sources no longer hash to this
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
