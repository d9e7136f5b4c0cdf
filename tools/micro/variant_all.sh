#!/bin/bash
# Experiment build of EVERY translation unit with extra flags: tools/micro/variant_all.sh <name> [flags ...]
#   -> tools/micro/libv_<name>.so      (UNITS="render pose" limits the flags to those units; the others are the in-tree objects)
set -e
cd "$(dirname "$0")/../.."
NAME=$1; shift
SRC=mipsfusion_amd/csrc
make -s -C $SRC >/dev/null
ALL="capi hashgrid elementwise render decoder decoder16 wgrad16 pose ro"
UNITS=${UNITS:-$ALL}
OBJS=""
for u in $ALL; do
  if [[ " $UNITS " == *" $u "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -w -Iinclude "$@" \
        -c $SRC/$u.hip -o tools/micro/va_${NAME}_$u.o 2>&1 | grep -v "not a recognized feature" || true &
    OBJS="$OBJS tools/micro/va_${NAME}_$u.o"
  else OBJS="$OBJS $SRC/$u.o"; fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o tools/micro/libv_$NAME.so
rm -f tools/micro/va_${NAME}_*.o
echo built tools/micro/libv_$NAME.so
