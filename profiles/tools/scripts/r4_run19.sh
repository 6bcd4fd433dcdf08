cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -5 ) 2>&1 | tee gpurun_out/r4_gpu_tests_final.log
bash profiles/tools/fwd_traffic.sh r4 cvt_native_K1 cvt_c3scale_K2 cvt_fvt10_K8 cv_native_K1 cv_c3scale_K2 cv_fvt10_K8 2>&1 | tail -7
timeout 900 python bench.py > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err; python profiles/tools/bench_digest.py gpurun_out/r4_bench.json | cut -c1-230
