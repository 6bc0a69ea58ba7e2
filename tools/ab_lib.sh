#!/bin/bash
# A/B of two builds of the library on one box: libpcc_old.so (tools/build_old_lib.sh) against the current one, alternating
# processes; prints the medians of tools/time_call.py.   usage: bash tools/ab_lib.sh [what]
cd "$(dirname "$0")/.."
W=${1:-match_cost}
for rep in 1 2 3; do
  PCC_LIB_OVERRIDE=$PWD/pointcloudcounterfactual_amd/lib/libpcc_old.so python3 tools/time_call.py $W old 2>&1 | grep -v amdgpu.ids
  python3 tools/time_call.py $W new 2>&1 | grep -v amdgpu.ids
done
