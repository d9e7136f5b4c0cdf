#!/bin/bash
# GPU box: tools/micro/sweep_cmd.sh <source-stem> <script.py> <kernel-grep> "<flags A>" "<flags B>" ...
# like sweep_flags.sh, but profiles an arbitrary python script (tools/micro/go_probe.py, ro_probe.py) instead of bench.py
STEM=$1; SCRIPT=$2; PAT=$3; shift 3
cd $GRAFT_REPO_ROOT/mipsfusion_amd/csrc
n=0
for f in "$@"; do
  n=$((n+1))
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -w -I../../include $f -c $STEM.hip -o $STEM.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC capi.o hashgrid.o elementwise.o render.o decoder.o decoder16.o wgrad16.o pose.o ro.o -o ../libmipsf_hip.so
  cd $GRAFT_REPO_ROOT; tools/profile_cmd.sh sc_$n $SCRIPT > /dev/null 2>&1
  echo "[$f] $(python tools/show_stats.py gpurun_out/prof_sc_$n/sc_${n}_kernel_stats.csv 30 | grep -i "$PAT" | cut -c13-40,64-100 | tr '\n' '|') $(python3 - gpurun_out/prof_sc_$n/run.log <<'PY'
import json, sys
out = ""
for line in open(sys.argv[1], errors="replace"):
    line = line.strip()
    if line.startswith("{") and '"ms_per_step"' in line:
        try:
            out = "step %.4f ms" % json.loads(line)["ms_per_step"]
        except Exception:
            pass
    elif " ms" in line and len(line) < 160 and not line.startswith(("W2026", "E2026", "I2026", "[bench")):
        out = line
print(out)
PY
)"
  cd $GRAFT_REPO_ROOT/mipsfusion_amd/csrc
done
