import json, sys
d = json.load(open(sys.argv[1]))
print("ms/step", d["ms_per_step"], "value", f'{d["value"]:.4g}', d["unit"], "| fwd-only ms", d["forward_only"]["ms"])
tot = 0
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["avg_ms"] * kv[1]["launches"]):
    per_step = v["avg_ms"] * v["launches"] / d["steps"]
    tot += per_step
    print(f'  {k:20s} {v["avg_ms"]*1000:8.1f} us x{v["launches"]:3d}  {v["bound"]:4s} frac {v["frac"]:.4f}  ({v["achieved"]} {v["unit"]})')
print(f"  sum of kernels per step: {tot*1000:.0f} us; unaccounted: {(d['ms_per_step']-tot)*1000:.0f} us")
for k in ("frame", "cpu_baseline", "roofline"):
    if k in d: print(k, d[k])
