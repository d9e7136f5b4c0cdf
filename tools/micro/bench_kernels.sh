#!/bin/bash
# GPU box: tools/micro/bench_kernels.sh [bench args]: step time + per-kernel avg_ms of one bench run, one line
python bench.py --steps 20 --warmup 5 --cpu-rays 0 --seq-frames 0 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('step', d['ms_per_step'], ' '.join('%s=%.1f' % (k, 1e3 * v['avg_ms']) for k, v in d['kernels'].items() if v.get('avg_ms')))"
