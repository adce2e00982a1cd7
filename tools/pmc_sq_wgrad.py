"""SQ counters of the three forms of conv4's weight gradient (tools/prof_sq_wgrad.sh): per-launch averages per kernel instantiation -> one JSON."""
import collections, csv, glob, json, os, sys
src, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "wgrad_bf16_dma_kernel<true, true" not in k:
            continue
        form = {"0": "dense", "1": "sparse operand compressed from dout", "2": "sparse operand from the pooled gradient"}[k.split("true, true, ")[1][0]]
        acc[form][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {form: {c: sum(v) / len(v) for c, v in sorted(cs.items())} | {"_launches": len(next(iter(cs.values())))} for form, cs in acc.items()}
res["_note"] = ("per-launch averages of wgrad_bf16_dma_kernel<FAST, GROUPED, SPARSE 0 / 1 / 2> at 256 -> 256, 32x32, 2304 images, 12 groups (tools/probe_wgrad_sparse.py); two "
                "rocprofv3 --pmc passes of 8 SQ counters each.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles.")
json.dump(res, open(out, "w"), indent=1)
for form, cs in res.items():
    if form.startswith("_"): continue
    print(form, {k: (round(v / 1e6, 1) if isinstance(v, float) else v) for k, v in cs.items()})
