cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -6 ) 2>&1 | tee gpurun_out/r4_gpu_tests_final.log
timeout 900 python bench.py > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err; python profiles/tools/bench_digest.py gpurun_out/r4_bench.json | cut -c1-260
bash profiles/run_rocprof.sh r4 > /dev/null 2>&1
bash profiles/run_rocprof_train.sh r4 > /dev/null 2>&1
for wl in fvt10_K8 c3scale_K2 native_K1; do
  ( cd /tmp; CV_ONE=$wl timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r4cv_$wl -o cv -- python $GRAFT_REPO_ROOT/profiles/tools/cv_bwd_form_ab.py > /dev/null 2>&1 )
done
bash profiles/tools/fwd_traffic.sh r4 raster_c3 raster_c2 cv_native_K1 ptf_2_views ptf_10_views 2>&1 | tail -6
bash profiles/run_rocprof_encoder.sh r4 > /dev/null 2>&1
ls gpurun_out | head -40
du -sh gpurun_out
