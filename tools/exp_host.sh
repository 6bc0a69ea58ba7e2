#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/exp_host.log
: > $O
run() { timeout -k 10 180 env "$@" python3 tools/exp_host.py >> $O 2>&1 || echo "FAILED: $*" >> $O; }
run PCC_X=0
run PCC_AM_NOSPLIT=1
run PCC_AM_G2_FROM=99 PCC_AM_G4_FROM=99
grep -v amdgpu.ids $O
