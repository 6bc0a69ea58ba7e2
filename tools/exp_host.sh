#!/bin/bash
# host enqueue time vs GPU time of pcc_match_cost (tools/exp_host.py), with one and two lanes
cd "$(dirname "$0")/.."
O=gpurun_out/exp_host.log
: > $O
run() { timeout -k 10 180 env "$@" python3 tools/exp_host.py >> $O 2>&1 || echo "FAILED: $*" >> $O; }
run PCC_X=0
run PCC_AM_NOSPLIT=1
grep -v amdgpu.ids $O
