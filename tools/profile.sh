#!/bin/bash
# usage: tools/profile.sh <tag> [bench args...]   (run on the GPU box through gpurun)
# rocprofv3 kernel trace + stats of bench.py; only the small summaries are copied into gpurun_out/.
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_$TAG
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$TAG -o $TAG -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-rays 0 --no-frame-estimate "$@" > $OUT/bench_under_rocprof.log 2>&1
echo "rocprofv3 rc=$?"
find /tmp/rp_$TAG -name "*stats*.csv" -exec cp {} $OUT/ \;
ls -la $OUT
