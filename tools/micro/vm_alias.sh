#!/bin/bash
# one process alone, then two at once on the same device
cd "$(dirname "$0")"
S=${1:-8}
./vm_alias 0 3
./vm_alias 0 $S & ./vm_alias 1 $S & wait
