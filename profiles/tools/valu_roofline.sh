#!/bin/bash
# VALU-issue roofline of the two blend kernels: class counters + busy cycles + isolated durations of the
# forward and the training benches, every counter set in its own rocprofv3 run (kernel trace only beside it), one raster stream
# so that launches do not overlap.   Usage (GPU box, repo root): profiles/tools/valu_roofline.sh <tag>
export TMPDIR=/tmp
export FREESPLAT_RASTER_STREAMS=1
TAG=$1
OUT=gpurun_out/valu_$TAG
mkdir -p $OUT
FWD="python bench.py --steps 4 --warmup 2 --views 8 --sections raster --no-graph --no-cpu-baseline --no-profile --min-time 0"
TRN="python bench.py --steps 3 --warmup 1 --views 8 --sections raster --no-graph --no-cpu-baseline --no-profile --min-time 0 --mode train"
SET_A="SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU"
SET_B="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE"
for M in fwd trn; do
  if [ $M = fwd ]; then B=$FWD; else B=$TRN; fi
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/${M}_trace -o x --output-format csv -- $B > $OUT/${M}_trace.log 2>&1
  timeout 300 rocprofv3 --pmc $SET_A --kernel-trace -d $OUT/${M}_a -o x --output-format csv -- $B > $OUT/${M}_a.log 2>&1
  timeout 300 rocprofv3 --pmc $SET_B --kernel-trace -d $OUT/${M}_b -o x --output-format csv -- $B > $OUT/${M}_b.log 2>&1
done
profiles/tools/_bin/valu_issue_rate > $OUT/valu_issue_rate.txt 2>&1
python profiles/tools/valu_roofline.py $OUT > $OUT/valu_roofline.json 2> $OUT/valu_roofline.err
tail -c 3000 $OUT/valu_roofline.json
