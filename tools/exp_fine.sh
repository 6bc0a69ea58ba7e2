#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/exp_fine.log
: > $O
run() { timeout -k 10 120 env "$@" python3 tools/exp_host.py 2>&1 | grep "40 calls" >> $O || echo "FAILED: $*" >> $O; }
run PCC_X=0
run PCC_AM_FINE=0
run PCC_AM_FINE_CA2=0
run PCC_AM_FINE=0 PCC_AM_G2_FROM=99 PCC_AM_G4_FROM=99
run PCC_AM_G2_FROM=3 PCC_AM_G4_FROM=4
run PCC_AM_G2_FROM=3 PCC_AM_G4_FROM=5
run PCC_AM_G2_FROM=4 PCC_AM_G4_FROM=6
cat $O
