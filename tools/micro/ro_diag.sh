#!/bin/bash
# The particle kernel's lanes fault beside a second process, build by build, on ONE box:
#   tools/micro/ro_diag.sh <launches> <lib> [<lib> ...]      (lib: a tools/micro/libv_*.so name, or "product")
#   builds: tools/micro/variant.sh ropk<k> ro -DMIPSF_RO_PACKED=<k> -DMIPSF_KEEP_PACKED_FP32   (k: see ro.hip)
cd "$(dirname "$0")/../.."
N=$1; shift
for lib in "$@"; do
  if [ $lib = product ]; then unset MIPSF_LIB; else export MIPSF_LIB=$PWD/tools/micro/$lib; fi
  echo "== $lib"
  timeout 300 python tools/dbg_ro_lanes.py $N ${LOADFLAG---load} --beside 2>&1 | grep -v amdgpu.ids | grep "launches, stage\|component" | cut -c1-250 | head -${SHOW:-4}
done
