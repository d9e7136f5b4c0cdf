#!/bin/bash
# GPU box: tools/micro/sweep_define.sh <source-stem> <MACRO> <kernel-grep> <value>...
# rebuilds one object with -D<MACRO>=<value>, relinks the library in the box's copy of the repo, runs bench.py under
# rocprofv3 and prints the average duration of the kernels matching <kernel-grep>.  (Tuning on the real mapping
# workload: the synthetic points of tools/bench_hashgrid.py ranked two scatter variants the wrong way round.)
STEM=$1; MACRO=$2; PAT=$3; shift 3
cd $GRAFT_REPO_ROOT/mipsfusion_amd/csrc
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -w -I../../include -D$MACRO=$v -c $STEM.hip -o $STEM.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC capi.o hashgrid.o elementwise.o render.o decoder.o decoder16.o wgrad16.o pose.o ro.o -o ../libmipsf_hip.so
  cd $GRAFT_REPO_ROOT; tools/profile.sh sw_$v --steps 20 --warmup 5 > /dev/null 2>&1
  echo "$MACRO=$v: $(python tools/show_stats.py gpurun_out/prof_sw_$v/sw_${v}_kernel_stats.csv 14 | grep -i "$PAT" | awk '{print $2, $(NF-3), "us;"}' | tr '\n' ' ')"
  cd $GRAFT_REPO_ROOT/mipsfusion_amd/csrc
done
