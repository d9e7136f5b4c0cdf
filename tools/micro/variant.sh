#!/bin/bash
# Experiment build of ONE translation unit: tools/micro/variant.sh <name> <stem> [-DFLAG ...]
#   compiles mipsfusion_amd/csrc/<stem>.hip with the flags and links it with the in-tree objects of the other units
#   -> tools/micro/libv_<name>.so (same C ABI; run with MIPSF_LIB=$PWD/tools/micro/libv_<name>.so)
set -e
cd "$(dirname "$0")/../.."
NAME=$1; STEM=$2; shift 2
SRC=mipsfusion_amd/csrc
make -s -C $SRC >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -w -Iinclude "$@" \
    -c $SRC/$STEM.hip -o tools/micro/v_$NAME.o
OBJS=""
for u in capi hashgrid elementwise render decoder decoder16 wgrad16 pose ro; do
  if [ $u = $STEM ]; then OBJS="$OBJS tools/micro/v_$NAME.o"; else OBJS="$OBJS $SRC/$u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o tools/micro/libv_$NAME.so
rm -f tools/micro/v_$NAME.o
echo built tools/micro/libv_$NAME.so
