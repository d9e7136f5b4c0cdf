#!/bin/bash
# GPU box: tools/micro/sweep_flags.sh <source-stem> <kernel-grep> "<flags A>" "<flags B>" ...
# like sweep_define.sh, but every variant is a whole set of -D flags
STEM=$1; PAT=$2; shift 2
cd $GRAFT_REPO_ROOT/mipsfusion_amd/csrc
n=0
for f in "$@"; do
  n=$((n+1))
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -w -I../../include $f -c $STEM.hip -o $STEM.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC capi.o hashgrid.o elementwise.o render.o decoder.o pose.o ro.o -o ../libmipsf_hip.so
  cd $GRAFT_REPO_ROOT; tools/profile.sh sf_$n --steps 20 --warmup 5 > /dev/null 2>&1
  echo "[$f] $(python tools/show_stats.py gpurun_out/prof_sf_$n/sf_${n}_kernel_stats.csv 30 | grep -i "$PAT" | awk '{print substr($1,1,30) substr($2,1,26), $(NF-3)}' | tr '\n' '|')"
  cd $GRAFT_REPO_ROOT/mipsfusion_amd/csrc
done
