#!/bin/bash
# GPU box: tools/micro/two_rank_loop.sh <n>  -- bench.py --gpus 2 (gloo, both ranks on this GPU) n times; prints the particle-split check
cd $GRAFT_REPO_ROOT
for i in $(seq $1); do
  MIPSF_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29500 + i)) bench.py --gpus 2 --steps 10 --warmup 5 --setup-iters 10 --cpu-rays 0 --no-frame-estimate 2>gpurun_out/two_rank_err.log | grep "^{" | tail -1 | python tools/micro/two_rank_line.py $i || tail -5 gpurun_out/two_rank_err.log
done
