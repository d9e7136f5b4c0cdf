#!/bin/bash
# usage: tools/pmc_cmd.sh <tag> <script.py> [args...]   (GPU box)
# separate rocprofv3 --pmc passes (kernel-trace only) over an arbitrary python script; per-kernel means -> gpurun_out/pmc_<tag>/
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
IFS=';' read -ra SETS <<< "${PMC_SETS:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU;SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_SMEM SQ_IFETCH}"
for SET in "${SETS[@]}"; do
  i=$((i+1))
  rm -rf /tmp/pmc_${TAG}_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pmc_${TAG}_$i -o p -- python3 $GRAFT_REPO_ROOT/"$@" > $OUT/pass$i.log 2>&1
  echo "pass $i ($SET) rc=$?"
  f=$(find /tmp/pmc_${TAG}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $f > $OUT/pass${i}_summary.txt
done
cat $OUT/pass*_summary.txt
