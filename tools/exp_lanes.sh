#!/bin/bash
# A/B runs of tools/exp_lanes.py (one process per switch setting: the switches are read once per process)
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/exp_lanes.log
: > $O
run() { timeout -k 10 120 env "$@" python3 tools/exp_lanes.py recon >> $O 2>&1 || echo "FAILED: $*" >> $O; }
run PCC_AM_NOSPLIT=1
run PCC_AM_LANES=2
run PCC_AM_LANES=3
run PCC_AM_LANES=4
run PCC_AM_NOCULL=1 PCC_EXP_GRAPH=0
timeout -k 10 120 env PCC_EXP_GRAPH=0 python3 tools/exp_lanes.py uniform >> $O 2>&1
timeout -k 10 120 env PCC_EXP_GRAPH=0 PCC_AM_NOCULL=1 python3 tools/exp_lanes.py uniform >> $O 2>&1
timeout -k 10 120 env PCC_AM_DEBUG=2 PCC_EXP_GRAPH=0 python3 tools/exp_lanes.py recon 2> gpurun_out/exp_stamps.log | tail -2 >> $O
cat $O
