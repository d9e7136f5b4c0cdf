#!/bin/bash
# usage: tools/profile_cmd.sh <tag> <script.py> [args...]   (GPU box) -- rocprofv3 kernel stats of any python script
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_$TAG
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$TAG -o $TAG -- python3 $GRAFT_REPO_ROOT/"$@" > $OUT/run.log 2>&1
echo "rocprofv3 rc=$?"
find /tmp/rp_$TAG -name "*kernel_stats.csv" -exec cp {} $OUT/ \;
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")
PY
