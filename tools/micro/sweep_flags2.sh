#!/bin/bash
# GPU box: tools/micro/sweep_flags2.sh <source-stem> <kernel-grep> "<flags A>" "<flags B>" ...   (several -D per variant)
STEM=$1; PAT=$2; shift 2
cd $GRAFT_REPO_ROOT/mipsfusion_amd/csrc
n=0
for f in "$@"; do
  n=$((n+1))
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -w -I../../include $f -c $STEM.hip -o $STEM.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC capi.o hashgrid.o elementwise.o render.o decoder.o decoder16.o wgrad16.o pose.o ro.o -o ../libmipsf_hip.so
  cd $GRAFT_REPO_ROOT; tools/profile.sh sf_$n --steps 20 --warmup 5 > /dev/null 2>&1
  echo "[$f]"; python tools/show_stats.py gpurun_out/prof_sf_$n/sf_${n}_kernel_stats.csv 30 | grep -i "$PAT" | cut -c1-50,65-100
  grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_sf_$n/bench_under_rocprof.log | head -1
  cd $GRAFT_REPO_ROOT/mipsfusion_amd/csrc
done
