lscpu | grep -i "numa\|model name\|socket" | head -12
run() { python - "$1" <<'PY'
import os, sys, json, subprocess
n = int(sys.argv[1])
if n > 0:
    cpus = sorted(os.sched_getaffinity(0))[:n]
    os.sched_setaffinity(0, cpus)
out = subprocess.run([sys.executable, "tools/run_sequence.py", "--frames", "31", "--graph", "--sampler", "reference"], capture_output=True, text=True).stdout.strip().split("\n")[-1]
import numpy as np
d = json.loads(out); f = np.array(d["frame_ms_all"][1:]); ba = f[f > 10]
print("affinity %3d cpus: mean(excl first) %.2f  BA-frame median %.2f  wait %.2f  host %s" % (n, f.mean(), np.median(ba), d["producer_wait_ms_mean"], d["producer_host_ms_per_frame"]))
PY
}
for rep in 1 2 3; do run 0; run 16; run 32; done
