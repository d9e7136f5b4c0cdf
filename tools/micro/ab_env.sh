#!/bin/bash
# GPU box: tools/micro/ab_env.sh <rounds> "<ENV=1>"   -- graphed headline step with and without an environment switch
R=$1; ENVV=$2
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/var
for r in $(seq $R); do
  for v in base env; do
    if [ $v = base ]; then python bench.py --steps 40 --warmup 10 --cpu-rays 0 --no-frame-estimate --no-variants > gpurun_out/var/ab_$v.json 2>gpurun_out/var/ab_$v.err
    else env $ENVV python bench.py --steps 40 --warmup 10 --cpu-rays 0 --no-frame-estimate --no-variants > gpurun_out/var/ab_$v.json 2>gpurun_out/var/ab_$v.err; fi
    python tools/micro/show_variants.py ab_$v 2>/dev/null || python - $v <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/var/ab_{sys.argv[1]}.json")); print(sys.argv[1], d["ms_per_step"])
PY
  done
done
