cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -6
run() { python tools/run_sequence.py --frames 31 --graph --sampler reference 2>/dev/null | tail -1 | python -c "
import sys, json, numpy as np
d=json.loads(sys.stdin.read()); f=np.array(d['frame_ms_all'][1:]); ba=f[f>10]; tr=f[f<=10]
print('$1: mean(excl first) %.2f  BA-frame median %.2f  tracking median %.2f  wait %.2f  host %s' % (f.mean(), np.median(ba), np.median(tr), d['producer_wait_ms_mean'], d['producer_host_ms_per_frame']))"; }
for rep in 1 2; do
run default
OMP_WAIT_POLICY=PASSIVE GOMP_SPINCOUNT=0 run passive
OMP_NUM_THREADS=4 run omp4
OMP_NUM_THREADS=4 OMP_WAIT_POLICY=PASSIVE GOMP_SPINCOUNT=0 run omp4_passive
done
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -6
