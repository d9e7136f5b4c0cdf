#!/bin/bash
# GPU box: tools/micro/run_variants.sh <kernel-grep> <name>...   ("base" = the in-tree library)
# runs the headline bench once per experiment library and prints ms/step + the matching per-kernel times
PAT=$1; shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/var
for v in "$@"; do
  if [ $v = base ]; then unset MIPSF_LIB; else export MIPSF_LIB=$GRAFT_REPO_ROOT/tools/micro/libv_$v.so; fi
  python bench.py --steps 20 --warmup 5 --cpu-rays 0 --no-frame-estimate ${BENCH_ARGS} > gpurun_out/var/$v.json 2> gpurun_out/var/$v.err || { echo "$v: FAILED"; tail -3 gpurun_out/var/$v.err; continue; }
  python - "$v" "$PAT" <<'PY'
import json, sys, re
v, pat = sys.argv[1], sys.argv[2]
d = json.load(open(f"gpurun_out/var/{v}.json"))
ks = {k: round(x["avg_ms"] * 1e3, 1) for k, x in d["kernels"].items() if re.search(pat, k)}
print(f"{v:24s} ms/step {d['ms_per_step']:.4f} eager {d['eager_ms_per_step']:.4f}  {ks}")
PY
done
