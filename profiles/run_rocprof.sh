#!/bin/bash
# Usage (on the GPU box, from the repo root): profiles/run_rocprof.sh <tag> [bench args...]
# Kernel trace + stats, the two HBM-traffic PMC passes and two SQ passes, each in its own run.
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
B="python bench.py --steps 5 --warmup 2 --sections raster --no-graph --no-cpu-baseline --no-profile --min-time 0 $*"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o c3 --output-format csv -- $B > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o c3 --output-format csv -- $B > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o c3 --output-format csv -- $B > $OUT/write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace -d $OUT/sq1 -o c3 --output-format csv -- $B > $OUT/sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --kernel-trace -d $OUT/sq2 -o c3 --output-format csv -- $B > $OUT/sq2.log 2>&1
ls $OUT
