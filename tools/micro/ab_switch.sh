for rep in 1 2; do
for si in 5e-3 1e-4 2e-5; do
  MIPSF_SWITCH_INTERVAL=$si python tools/run_sequence.py --frames 31 --graph --sampler reference 2>/dev/null | tail -1 | python -c "
import sys, json, numpy as np
d=json.loads(sys.stdin.read()); f=np.array(d['frame_ms_all'][1:]); ba=f[f>10]; tr=f[f<=10]
print('switch $si: mean(excl first) %.2f  BA-frame median %.2f  tracking median %.2f  wait %.2f  host %s' % (f.mean(), np.median(ba), np.median(tr), d['producer_wait_ms_mean'], d['producer_host_ms_per_frame']))"
done; done
