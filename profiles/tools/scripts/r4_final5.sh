cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
bash profiles/tools/fwd_traffic.sh r4 ptf_2_views ptf_10_views 2>&1 | grep ptf_
( time timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -3 ) 2>&1 | tee gpurun_out/r4_gpu_tests_final.log
timeout 900 python bench.py > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err; python profiles/tools/bench_digest.py gpurun_out/r4_bench.json | cut -c1-170
bash profiles/run_rocprof_encoder.sh r4 > /dev/null 2>&1
