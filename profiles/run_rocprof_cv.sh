#!/bin/bash
# PMC passes for the cost-volume kernel (MFMA utilisation): profiles/run_rocprof_cv.sh <tag>
export TMPDIR=/tmp
OUT=gpurun_out/prof_$1_cv
mkdir -p $OUT
B="python -c \"import bench_encoder as b, torch; b.bench_cost_volume(torch.device('cuda:0'), 5, 2)\""
eval rocprofv3 --kernel-trace --stats -d $OUT/trace -o cv --output-format csv -- $B > $OUT/trace.log 2>&1
eval rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d $OUT/sq1 -o cv --output-format csv -- $B > $OUT/sq1.log 2>&1
eval rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $OUT/sq2 -o cv --output-format csv -- $B > $OUT/sq2.log 2>&1
eval rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o cv --output-format csv -- $B > $OUT/fetch.log 2>&1
ls $OUT
