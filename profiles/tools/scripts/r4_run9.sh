cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 ) 2>&1 | tee gpurun_out/r4_gpu_tests_full.log
python bench.py --sections c5 --no-graph 2>/dev/null > gpurun_out/r4_c5.json; python profiles/tools/bench_digest.py gpurun_out/r4_c5.json
