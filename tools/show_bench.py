"""Compact view of a bench.py JSON line: python tools/show_bench.py <file>"""
import json
import sys

d = json.load(open(sys.argv[1]))
print("ms/step", d["ms_per_step"], "value", f'{d["value"]:.4g}', d["unit"], "| fwd-only ms", d["forward_only"]["ms"], "| n_gpus", d["n_gpus"])
tot = 0
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["avg_ms"] * kv[1]["launches"]):
    per_step = v["avg_ms"] * v["launches"] / d["steps"]
    tot += per_step
    print(f'  {k:20s} {v["avg_ms"]*1000:8.1f} us x{v["launches"]:3d}  {v["bound"]:4s} frac {v["frac"]:.4f}  ({v["achieved"]} {v["unit"]})')
print(f"  sum of kernels per step: {tot*1000:.0f} us; unaccounted: {(d['ms_per_step']-tot)*1000:.0f} us")
for k, v in (d.get("variants") or {}).items():
    print(f"  variant {k}: " + ", ".join(f"{a} {v[a]}" for a in ("ms_per_step", "ms_per_step_indices_predrawn") if a in v))
    if "breakdown_ms_per_step_serialised" in v:
        print("     breakdown:", v["breakdown_ms_per_step_serialised"])
f = d.get("frame") or {}
print("frame:", {k: v for k, v in f.items() if not isinstance(v, (dict, list, str))})
for key in ("measured_sequence_fast_mode", "config3_multi_submap"):
    if key in f:
        print(f"  {key}:", {a: f[key].get(a) for a in ("ms_per_frame_mean", "ms_per_frame_median", "ms_per_frame_mean_without_switch_frames", "ate_rmse_m")})
if "cpu_baseline" in d:
    c = d["cpu_baseline"]
    print("cpu_baseline:", {a: c[a] for a in ("value", "cores", "kind", "s_per_iter")}, c["gpu_vs_oracle_same_batch"])
r = d.get("roofline") or {}
print("roofline:", {a: r.get(a) for a in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic")})
if "multi_gpu" in d:
    print("multi_gpu:", json.dumps(d["multi_gpu"])[:600])
