#!/bin/bash
# which of the mapping step's kernels, run by a SECOND process, makes the packed build of ro_particles_kernel fault:
#   tools/micro/pk_neigh.sh <seconds> <regex> [<regex> ...]     ("all" = the whole step)
cd "$(dirname "$0")"
S=$1; shift
PK=$PWD/libv_ropk1.so
for re in "$@"; do
  if [ "$re" = all ]; then python ../ba_load.py --seconds $((S + 25)) > /tmp/ba_load.log 2>&1 & NB=$!
  else python ../ba_load.py --seconds $((S + 25)) --only "$re" > /tmp/ba_load.log 2>&1 & NB=$!; fi
  sleep 22
  echo "-- neighbour: $re"; ./pk_lanes lib:$PK $S | cut -c1-200
  wait $NB; tail -1 /tmp/ba_load.log | cut -c1-300
done
