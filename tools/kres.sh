#!/bin/bash
# kernel resource summary of one .hip file of the product: name, VGPRs, AGPRs, spills, scratch, LDS   (tools/kres.sh decoder16.hip [extra flags])
HERE="$(cd "$(dirname "$0")" && pwd)"
cd "$HERE/../mipsfusion_amd/csrc" || exit 1
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -I../../include "$@" -c "$f" -o /tmp/kres_$$.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 "$HERE/kres.py"
rm -f /tmp/kres_$$.o
