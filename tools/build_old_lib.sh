#!/bin/bash
# Build the library of a given commit (default HEAD) as pointcloudcounterfactual_amd/lib/libpcc_old.so for A/B runs in
# separate processes on one box (PCC_LIB_OVERRIDE; tools/ab_lib.sh).
set -e
REV=${1:-HEAD}
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
git -C "$R" archive "$REV" pointcloudcounterfactual_amd/csrc include | tar -x -C "$T"
make -C "$T/pointcloudcounterfactual_amd/csrc" -j4 OUT="$R/pointcloudcounterfactual_amd/lib/libpcc_old.so" > /dev/null
rm -rf "$T"
ls -la "$R/pointcloudcounterfactual_amd/lib/libpcc_old.so"
