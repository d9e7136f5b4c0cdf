#!/bin/bash
# Build a PRIVATE experiment copy of the library: tools/micro/build_variant.sh <suffix> [-D flags...]
#   -> tools/micro/libmipsf_<suffix>.so   (select it with MIPSF_LIB_VARIANT=<suffix> in tools/bench_decoder.py)
set -e
cd "$(dirname "$0")/../.."
SRC=mipsfusion_amd/csrc
SUF=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -w "$@" \
    -Iinclude -shared $SRC/capi.hip $SRC/hashgrid.hip $SRC/elementwise.hip $SRC/render.hip $SRC/decoder.hip $SRC/pose.hip $SRC/ro.hip \
    -o tools/micro/libmipsf_$SUF.so
echo built tools/micro/libmipsf_$SUF.so
