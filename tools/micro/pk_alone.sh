#!/bin/bash
# tools/micro/pk_alone.sh [seconds]: the stand-alone reproducer of DESIGN.md 4h -- nothing but this program: a second process runs
# its `neighbour` kernels (kind: see pk_lanes.hip), the first one the victim kernel with the crossed packed add (asm), with the
# one-instruction fix (asmfix), as hipcc packs this file's source (packed), with single multiplies (single)
cd "$(dirname "$0")"
S=${1:-8}
for kind in ${KINDS:-11 10 0}; do
  echo "-- neighbour process: kind $kind"
  ./pk_lanes neighbour $((4 * S + 4)) kind:$kind & NB=$!
  sleep 1
  for m in asm asmfix packed single; do ./pk_lanes $m $S | cut -c1-190; done
  wait $NB
done
echo "-- alone"; ./pk_lanes asm $S | cut -c1-190
