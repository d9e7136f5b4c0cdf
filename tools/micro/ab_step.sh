#!/bin/bash
# GPU box: tools/micro/ab_step.sh <rounds> <name>...  -- graphed headline step time per library, alternating ("base" = in-tree)
R=$1; shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/var
for r in $(seq $R); do
for v in "$@"; do
  if [ $v = base ]; then unset MIPSF_LIB; else export MIPSF_LIB=$GRAFT_REPO_ROOT/tools/micro/libv_$v.so; fi
  python bench.py --steps 40 --warmup 10 --cpu-rays 0 --no-frame-estimate --no-variants > gpurun_out/var/ab_$v.json 2> gpurun_out/var/ab_$v.err || { echo "$v FAILED"; continue; }
  python - $v <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/var/ab_{sys.argv[1]}.json"))
print(f"{sys.argv[1]:12s} ms/step {d['ms_per_step']:.4f}")
PY
done; done
