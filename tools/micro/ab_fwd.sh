#!/bin/bash
# GPU box: tools/micro/ab_fwd.sh <rounds> <name>...   -- the forward probe once per round per library ("base" = in-tree)
R=$1; shift
for r in $(seq $R); do
for v in "$@"; do
  if [ $v = base ]; then unset MIPSF_LIB; else export MIPSF_LIB=$GRAFT_REPO_ROOT/tools/micro/libv_$v.so; fi
  python tools/micro/fwd_probe.py 2>&1 | grep "save=" | tr '\n' ' '; echo
done; done
