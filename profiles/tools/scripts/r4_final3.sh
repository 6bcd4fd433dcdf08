cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -6 ) 2>&1 | tee gpurun_out/r4_gpu_tests_final.log
timeout 900 python bench.py > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err; python profiles/tools/bench_digest.py gpurun_out/r4_bench.json | cut -c1-260
bash profiles/run_rocprof_encoder.sh r4 > /dev/null 2>&1
( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r4_ptf_train -o x -- python $GRAFT_REPO_ROOT/profiles/tools/ptf_train_prof.py > /dev/null 2>&1 )
ls gpurun_out | head -40
