#!/bin/bash
# rocprofv3 passes for the encoder-side kernels: profiles/run_rocprof_encoder.sh <tag>
# cost volume (native 96x128, K=1): kernel trace + MFMA counters + HBM traffic; PTF (2-view fold @ 384x512): kernel
# trace + HBM traffic.  Every --pmc set in its own run (with --kernel-trace only).  Summaries: profiles/summarize_encoder.py
export TMPDIR=/tmp
TAG=$1
for W in cv ptf; do
  OUT=gpurun_out/prof_${TAG}_$W
  mkdir -p $OUT
  if [ $W = cv ]; then B="python -c \"import bench_encoder as b, torch; b.bench_cost_volume(torch.device('cuda:0'), 5, 2, cpu=False)\""
  else B="python -c \"import bench_encoder as b, torch; b.bench_ptf(torch.device('cuda:0'), 5, 2, cpu=False)\""; fi
  eval timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o x --output-format csv -- $B > $OUT/trace.log 2>&1
  eval timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o x --output-format csv -- $B > $OUT/fetch.log 2>&1
  eval timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o x --output-format csv -- $B > $OUT/write.log 2>&1
  eval timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d $OUT/sq1 -o x --output-format csv -- $B > $OUT/sq1.log 2>&1
  eval timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $OUT/sq2 -o x --output-format csv -- $B > $OUT/sq2.log 2>&1
  python profiles/summarize_encoder.py $OUT ${TAG}_$W "$B"
done
