#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root).
#   tools/profile_bench.sh <tag>   -> gpurun_out/<tag>_{stats,pmc_fetch,pmc_write,pmc_sq}/...
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --reps 1 --no-cpu-baseline --no-extras --no-noskip ${BENCH_EXTRA:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $R/bench.py $ARGS > $OUT/${TAG}_stats.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/${TAG}_pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- python3 $R/bench.py $ARGS > $OUT/${TAG}_pmc_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/${TAG}_pmc_sq -- python3 $R/bench.py $ARGS > $OUT/${TAG}_pmc_sq.log 2>&1 || exit 1
# VALU / transcendental split (an exponential occupies the vector pipe for 4 issue slots): its own pass
rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 --output-format csv -d $OUT/${TAG}_pmc_trans -- python3 $R/bench.py $ARGS > $OUT/${TAG}_pmc_trans.log 2>&1 || echo "no TRANS counters on this box"
# ---- secondary rows (k-NN, gathers and their backward, pooling, auction): the same bench WITH its extras ----
XARGS="--steps 3 --warmup 1 --reps 1 --no-cpu-baseline --no-noskip"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}x_stats -- python3 $R/bench.py $XARGS > $OUT/${TAG}x_stats.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}x_pmc_fetch -- python3 $R/bench.py $XARGS > $OUT/${TAG}x_pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}x_pmc_write -- python3 $R/bench.py $XARGS > $OUT/${TAG}x_pmc_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/${TAG}x_pmc_sq -- python3 $R/bench.py $XARGS > $OUT/${TAG}x_pmc_sq.log 2>&1 || echo "no SQ pass for the extras"
# ---- the no-skip probe (bench.py --phase-probe sets the am_nocull switch of pcc_test_hooks.h: every exact-zero skip off,
# algorithmic == executed work); the arming variable is inherited by the program ----
export PCC_TEST_HOOKS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}n_stats -- python3 $R/bench.py --phase-probe --steps 10 > $OUT/${TAG}n_stats.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $OUT/${TAG}n_pmc_sq -- python3 $R/bench.py --phase-probe --steps 10 > $OUT/${TAG}n_pmc_sq.log 2>&1 || echo "no SQ pass for the probe"
unset PCC_TEST_HOOKS
echo profiled
