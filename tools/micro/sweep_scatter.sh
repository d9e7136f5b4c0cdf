#!/bin/bash
# GPU box: tools/micro/sweep_scatter.sh "<flags A>" "<flags B>" ...: hashgrid.hip rebuilt with each flag set, bench.py under
# rocprofv3, one line per set: route / scan / zero / scatter / reduce kernel averages (us), their sum, and the step time
cd $GRAFT_REPO_ROOT/mipsfusion_amd/csrc
n=0
for f in "$@"; do
  n=$((n+1))
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -w -I../../include $f -c hashgrid.hip -o hashgrid.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC capi.o hashgrid.o elementwise.o render.o decoder.o decoder16.o wgrad16.o pose.o ro.o -o ../libmipsf_hip.so
  cd $GRAFT_REPO_ROOT; tools/profile_cmd.sh ss_$n bench.py --steps 20 --warmup 5 --cpu-rays 0 --seq-frames 0 > /dev/null 2>&1
  python3 - "$f" gpurun_out/prof_ss_$n <<'PY'
import csv, glob, json, sys
flags, d = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(glob.glob(d + "/*kernel_stats.csv")[0])))
def avg(pat):
    return sum(float(r["AverageNs"]) / 1e3 for r in rows if pat in r["Name"])
parts = [avg("scatter_route_kernel"), avg("scatter_scan_kernel"), avg("scatter_zero_kernel"), avg("hashgrid_scatter_kernel"), avg("hashgrid_scatter_reduce_kernel")]
step = ""
for line in open(d + "/run.log", errors="replace"):
    if line.startswith("{") and '"ms_per_step"' in line:
        step = "step %.4f ms" % json.loads(line)["ms_per_step"]
print("[%s] route %.1f scan %.1f zero %.1f scatter %.1f reduce %.1f = %.1f us  %s" % ((flags,) + tuple(parts) + (sum(parts), step)))
PY
  cd $GRAFT_REPO_ROOT/mipsfusion_amd/csrc
done
