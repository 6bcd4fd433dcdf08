#!/bin/bash
# HBM-traffic PMC passes of every forward workload (on the GPU box, from the repo root): profiles/tools/fwd_traffic.sh <tag> [workloads...]
export TMPDIR=/tmp
TAG=$1; shift
WL=${@:-raster_c3 raster_c2 raster_closeup cv_native_K1 cv_c3scale_K2 cv_fvt10_K8 cv_fvt10_K8_cl cvt_native_K1 cvt_c3scale_K2 cvt_fvt10_K8 ptf_2_views ptf_10_views ptf_3_views}
for W in $WL; do
  OUT=gpurun_out/traffic_$TAG/$W
  mkdir -p $OUT
  CALLS=$(python -c "import sys; sys.path.insert(0,'profiles/tools'); import fwd_traffic as f; print(f.WORKLOADS['$W'][1])")
  echo $CALLS > $OUT/calls.txt
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o x --output-format csv -- python profiles/tools/fwd_traffic.py run $W $CALLS > $OUT/fetch.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o x --output-format csv -- python profiles/tools/fwd_traffic.py run $W $CALLS > $OUT/write.log 2>&1
done
python profiles/tools/fwd_traffic.py summarize $TAG
