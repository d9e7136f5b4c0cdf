run() { python tools/run_sequence.py --frames 25 --graph --sampler reference 2>/dev/null | tail -1 | python -c "
import sys, json, numpy as np
d=json.loads(sys.stdin.read()); f=np.array(d['frame_ms_all'][1:]); ba=f[f>10]; tr=f[f<=10]
rg=np.array(d['ro_go_ms_all'][1:]); bam = f>10
print('$1: mean(excl first) %.2f  BA-frame median %.2f  tracking median %.2f | on BA frames: RO %.2f GO %.2f | host %s' % (f.mean(), np.median(ba), np.median(tr), np.median(rg[bam,0]), np.median(rg[bam,1]), d['producer_host_ms_per_frame']))"; }
MIPSF_SEQ_VERBOSE=1 run all_on
MIPSF_SEQ_VERBOSE=1 MIPSF_DIAG_STAGE_OFF=python run python_off
MIPSF_SEQ_VERBOSE=1 MIPSF_DIAG_STAGE_OFF=topk run topk_off
MIPSF_SEQ_VERBOSE=1 MIPSF_DIAG_STAGE_OFF=torch run torch_off
