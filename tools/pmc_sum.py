import csv, sys, glob, collections, os
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if os.environ.get("KFILTER", "wgrad_bf16") in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            print(d.split("/")[-1], k, "avg %.4g" % (sum(v) / len(v)), "n", len(v))
