#!/bin/bash
# The measurements DESIGN.md section 8 (round 3) quotes, in one run on the GPU box -> gpurun_out/r03_experiments.log
# (copied to profiles/r03_experiments.log).  Usage through gpurun: bash tools/run_experiments.sh
cd "$(dirname "$0")/.."
O=gpurun_out/r03_experiments.log
mkdir -p gpurun_out
: > $O
run() { echo "### $*" >> $O; timeout -k 10 300 "$@" 2>&1 | grep -v amdgpu.ids >> $O || echo "FAILED: $*" >> $O; echo >> $O; }
run ./tools/mfma_filter_bench
run python3 tools/exp_lane_alone.py
run python3 tools/exp_lanes.py match_cost
run python3 tools/ab.py 1 match_cost
run python3 tools/ab.py 1 step
run python3 tools/time_fine.py 8
PCC_AM_NORESIDENT=1 run python3 tools/time_fine.py 8
run python3 tools/time_graph_bwd.py
PCC_EDGE_SCATTER=1 run python3 tools/time_graph_bwd.py
run python3 tools/time_emd.py
run ./tools/issue_bench
run ./tools/mfma_coissue_bench
run python3 tools/time_knn.py 1
run python3 tools/ab_nn.py
run python3 tools/host_enqueue.py
tail -5 $O
