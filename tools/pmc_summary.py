"""Per-kernel mean of each counter in a rocprofv3 counter_collection.csv (only mipsf:: kernels), with the kernel's own mean
duration IN THAT PASS (End - Start timestamps of the same rows) and, where GRBM_GUI_ACTIVE was collected, the effective clock
= GRBM_GUI_ACTIVE / duration (MI355X_MICROARCH.md, DVFS give-back) -- printed both for the raw counter and for the counter
divided by the 8 XCDs rocprofv3 sums it over."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for r in rows:
    name = r.get("Kernel_Name", "")
    if "mipsf" not in name:
        continue
    short = re.sub(r"^void ", "", name).split("(")[0].replace("mipsf::", "")
    acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    try:
        dur[short][r.get("Dispatch_Id", len(dur[short]))] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    except (KeyError, ValueError):
        pass
for k in sorted(acc):
    parts = [f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(acc[k].items())]
    n = len(next(iter(acc[k].values())))
    d = sum(dur[k].values()) / len(dur[k]) if dur[k] else None
    if d:
        parts.append(f"dur_us={d:.1f}")
        g = acc[k].get("GRBM_GUI_ACTIVE")
        if g:
            g = sum(g) / len(g)
            parts.append(f"clk_GHz[GRBM/dur]={g / d * 1e-3:.2f} clk_GHz[GRBM/8/dur]={g / 8 / d * 1e-3:.2f}")
        w = acc[k].get("SQ_VALU_MFMA_BUSY_CYCLES")
        if w and sum(w) > 0:
            # cycles of MFMA issue per SIMD (1024 SIMDs) over the pass's own duration: busy share at 1.4 / 2.2 / 2.4 GHz
            per_simd = sum(w) / len(w) / 1024.0
            parts.append("mfma_busy_share@1.4/2.2/2.4GHz=" + "/".join(f"{per_simd / (d * 1e3 * f):.2f}" for f in (1.4, 2.2, 2.4)))
    print(f"{k:48s} n={n:3d} " + " ".join(parts))
