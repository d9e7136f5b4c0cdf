#!/bin/bash
# GPU box: tools/micro/sweep_route.sh "<flags A>" ...: hashgrid.hip rebuilt with each flag set, route_probe.py timed
cd $GRAFT_REPO_ROOT/mipsfusion_amd/csrc
for f in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -w -I../../include $f -c hashgrid.hip -o hashgrid.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC capi.o hashgrid.o elementwise.o render.o decoder.o decoder16.o wgrad16.o pose.o ro.o -o ../libmipsf_hip.so
  cd $GRAFT_REPO_ROOT
  echo "[$f] $(timeout 120 python tools/micro/route_probe.py 2>&1 | grep "^route")"
  cd $GRAFT_REPO_ROOT/mipsfusion_amd/csrc
done
