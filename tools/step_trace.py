"""Kernel sequence of one benchmark step (start offset, duration, gap to the previous kernel) from a rocprofv3 kernel trace.
usage (GPU box): rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_tr -o tr -- python3 bench.py ... ; python tools/step_trace.py /tmp/rp_tr"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "sample_rays_kernel" in r["Kernel_Name"]]
pairs = [(idx[k], idx[k + 1]) for k in range(len(idx) - 1)
         if any("adam_kernel" in rows[q]["Kernel_Name"] for q in range(idx[k], idx[k + 1]))]
a, b = pairs[-2]          # a full training step: sample placement ... Adam
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
tot_gap = 0.0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"]
    print("%8.1f us  dur %6.1f  gap %5.1f  %s" % ((s - t0) / 1000, (e - s) / 1000, (s - prev_end) / 1000, name[:100]))
    tot_gap += (s - prev_end) / 1000
    prev_end = e
print("step span %.1f us, gaps %.1f us, kernels %d" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1000, tot_gap, b - a))
