#!/bin/bash
# A/B of the frame loop's hand-over on bench.py's own 31-frame sequence: tools/micro/ab_seq.sh   (GPU box)
for i in 1 2; do for m in "MIPSF_SEQ_HOST_HANDOVER=1" "MIPSF_SEQ_NO_STAGE=1" "X=1"; do
env $m python3 bench.py --steps 20 --warmup 5 --no-variants --cpu-rays 0 --config3-frames 0 >/dev/null 2>&1
python3 -c "
import json; d=json.load(open('bench_detail.json'))['frame']['measured_sequence']
for s in ('reference','device'):
    r=d[s]; print('$m', s, r['ms_per_frame_mean'], r['ms_per_frame_median'], 'ro',r['ro_ms_mean'],'go',r['go_ms_mean'],'ba',r['ba_ms_per_round_median'],'wait',r['producer_wait_ms_mean'])
    print('   ', r['frame_ms_all'])"
done; done
