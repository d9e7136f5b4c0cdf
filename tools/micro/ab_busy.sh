python - <<'PY'
import sys; sys.path.insert(0, ".")
from mipsfusion_amd import hostcpu
b = hostcpu.cpu_busy_fractions(0.2)
nodes = hostcpu.numa_nodes()
for i, n in enumerate(nodes):
    cs = sorted(n)
    print("node", i, "busy cpus (>20%):", [c for c in cs if b.get(c, 0) > 0.2][:40], "mean busy %.3f" % (sum(b.get(c, 0) for c in cs) / len(cs)))
PY
for i in 1 2; do python bench.py --steps 20 --warmup 5 --cpu-rays 0 2>/dev/null | tail -1 | python -c "
import sys, json, numpy as np; d=json.loads(sys.stdin.read()); f=d['frame']; s=f['measured_sequence']['reference']; fm=np.array(s['frame_ms_all']); print(d['config']['host_cpus'], d['ms_per_step'], 'ref', s['ms_per_frame_mean'], 'wait', s['producer_wait_ms_mean'], s['producer_host_ms_per_frame'], 'dev', f['ms_per_frame_device_sampling'])"; done
