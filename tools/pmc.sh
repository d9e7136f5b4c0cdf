#!/bin/bash
# usage: tools/pmc.sh <tag> [script args...]   (GPU box) -- separate rocprofv3 --pmc passes (with --kernel-trace only),
# summarised per kernel; default program: bench.py, PMC_PASSES="1 3" restricts the passes
TAG=$1
shift
if [ $# -gt 0 ]; then PROG="$GRAFT_REPO_ROOT/$1"; shift; ARGS="$@"; else PROG="$GRAFT_REPO_ROOT/bench.py"; ARGS="--steps 5 --warmup 5 --stats-steps 10 --setup-iters 4 --cpu-rays 0 --no-frame-estimate --no-variants"; fi
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  if [ -n "$PMC_PASSES" ] && ! echo " $PMC_PASSES " | grep -q " $i "; then continue; fi
  rm -rf /tmp/pmc_${TAG}_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pmc_${TAG}_$i -o p -- python3 $PROG $ARGS > $OUT/pass$i.log 2>&1
  echo "pass $i ($SET) rc=$?"
  f=$(find /tmp/pmc_${TAG}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $f > $OUT/pass${i}_summary.txt
done
cat $OUT/pass*_summary.txt
