"""Per-kernel mean of each counter in a rocprofv3 counter_collection.csv (only mipsf:: kernels)."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r.get("Kernel_Name", "")
    if "mipsf" not in name:
        continue
    short = re.sub(r"^void ", "", name).split("(")[0].replace("mipsf::", "")
    acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    parts = [f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(acc[k].items())]
    n = len(next(iter(acc[k].values())))
    print(f"{k:48s} n={n:3d} " + " ".join(parts))
