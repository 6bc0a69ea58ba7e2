#!/bin/bash
# PMC counters of the c >= 4 k-NN kernels: tools/pmc_knn.sh <channels> <mode> -> gpurun_out/knn_pmc_<c>_<mode>.txt
R=$(pwd); OUT=$R/gpurun_out; C=$1; M=$2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/knn_pmc_${C}_${M} -- python3 $R/tools/run_knn.py $C $M > $OUT/knn_pmc_${C}_${M}.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/knn_pmc2_${C}_${M} -- python3 $R/tools/run_knn.py $C $M > $OUT/knn_pmc2_${C}_${M}.log 2>&1 || echo "second pass failed"
python3 $R/tools/pmc_table.py $OUT/knn_pmc_${C}_${M} $OUT/knn_pmc2_${C}_${M} knn_mfma > $OUT/knn_pmc_${C}_${M}.txt
