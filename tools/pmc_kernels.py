"""Per-kernel means of a rocprofv3 --pmc counter CSV: tools/pmc_kernels.py <dir> [name substring ...]
(the counter_collection CSV has one row per dispatch and counter; rows are grouped by kernel name and counter)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
subs = sys.argv[2:]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if subs and not any(s in name for s in subs):
            continue
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for name, cs in acc.items():
    out[name[:90]] = {c: sum(v) / len(v) for c, v in sorted(cs.items())}
    out[name[:90]]["dispatches"] = max(len(v) for v in cs.values())
print(json.dumps(out, indent=1))
