import csv, glob, collections
agg = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/pmc_sq*/**/*_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "kg::" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].replace("void ", "").split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{k:32s} {c:40s} n={len(v)} mean={sum(v)/len(v):.4g}")
