"""gpurun_out/pmc_<tag>/pass*_summary.txt -> profiles/pmc_latest.json (+ a readable copy of the summaries)."""
import json, os, re, sys, glob
tag = sys.argv[1]
src = f"gpurun_out/pmc_{tag}"
vals = {}
for f in sorted(glob.glob(os.path.join(src, "pass*_summary.txt"))):
    for line in open(f):
        m = re.match(r"(\S+?)(<[^>]*>)?\s+n=\s*(\d+)\s+(.*)", line)
        if not m:
            continue
        k = m.group(1)
        for kv in m.group(4).split():
            c, _, v = kv.partition("=")
            try:
                vals.setdefault(k, {})[c] = float(v)
            except ValueError:
                vals.setdefault(k, {})[c] = v
# map rocprof kernel names -> bench kernel groups (a group = one C-ABI call)
groups = {"hashgrid_fwd": ["hashgrid_fwd_kernel"],
          "hashgrid_bwd": ["scatter_zero_kernel", "scatter_route_kernel", "scatter_scan_kernel", "hashgrid_scatter_kernel",
                           "hashgrid_scatter_persist_kernel", "hashgrid_scatter_reduce_kernel"],
          "hashgrid_dx": ["hashgrid_dx_jac_kernel"],
          "decoder_fwd": ["decoder_fwd_kernel", "decoder_fwd_lds_kernel", "decoder16_fwd_kernel", "decoder16_fwd_lds_kernel"],
          "decoder_bwd_chain": ["decoder_bwd_lds_kernel", "decoder16_bwd_kernel", "decoder16_bwd_lds_kernel"],
          "decoder_wgrad": ["decoder_wgrad_kernel", "decoder_wgrad16_kernel", "decoder_wgrad_reduce_kernel"],
          "adam_step": ["adam_kernel", "adam_all_kernel"], "sample_rays": ["sample_rays_kernel"], "render_fwd": ["render_fwd_kernel", "render_train_kernel", "loss_finalize_kernel"],
          "render_bwd": ["render_bwd_kernel"], "rays_bwd": ["rays_bwd_kernel"]}
# FETCH_SIZE on gfx950 reports HALF the bytes of a 16-byte-per-lane streaming read (MI355X_MICROARCH.md, HBM section;
# calibrated here by adam_kernel: 72 MB reported for 144 MB of p/g/m/v reads).  Kernels whose reads are such streams get
# fetch x 2; the raw sum is kept next to it.  WRITE_SIZE is taken as reported (adam: 141 MB reported, 144 MB written).
WIDE_READS = {"decoder_fwd", "decoder_bwd_chain", "decoder_wgrad", "adam_step", "hashgrid_dx"}
traffic, traffic_raw, detail = {}, {}, {}
for g, ks in groups.items():
    fetch = sum(vals.get(k, {}).get("FETCH_SIZE", 0.0) for k in ks) * 1024.0
    write = sum(vals.get(k, {}).get("WRITE_SIZE", 0.0) for k in ks) * 1024.0
    traffic_raw[g] = round(fetch + write)
    traffic[g] = round((2.0 if g in WIDE_READS else 1.0) * fetch + write)
    detail[g] = {"fetch_reported": round(fetch), "write_reported": round(write), "fetch_x2": g in WIDE_READS}
out = {"source": f"rocprofv3 --pmc passes, tools/pmc.sh {tag}", "traffic_bytes_per_launch": traffic,
       "traffic_bytes_per_launch_uncorrected": traffic_raw, "traffic_detail": detail, "counters": vals}
os.makedirs("profiles", exist_ok=True)
json.dump(out, open("profiles/pmc_latest.json", "w"), indent=1)
with open(f"profiles/{tag[:3]}_{tag[3:]}_pmc_summary.txt" if tag.startswith("r0") and len(tag) == 4 else f"profiles/{tag}_pmc_summary.txt", "w") as o:
    for f in sorted(glob.glob(os.path.join(src, "pass*_summary.txt"))):
        o.write(open(f).read())
print(json.dumps(traffic, indent=1))
