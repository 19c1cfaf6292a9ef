"""Summarise rocprofv3 --pmc CSVs: average counter value per kernel."""
import collections, csv, glob, sys
root = sys.argv[1]
for d in sorted(glob.glob(f"{root}/pmc*/")):
    fs = glob.glob(f"{d}/*/*_counter_collection.csv")
    if not fs: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:34]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if k.startswith("dmr::"):
            print(k.ljust(36), {c: round(sum(x) / len(x) / 1e6, 2) for c, x in sorted(v.items())}, "(x1e6)")
