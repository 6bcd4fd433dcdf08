cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6_profiles
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r6_profiles/r6_gpu_tests.log 2>&1
tail -4 gpurun_out/r6_profiles/r6_gpu_tests.log
timeout 1200 python bench.py > gpurun_out/r6_profiles/r6_bench_stdout.log 2>&1
tail -1 gpurun_out/r6_profiles/r6_bench_stdout.log > gpurun_out/r6_profiles/r6_bench_headline.json
cp gpurun_out/bench_full.json gpurun_out/r6_profiles/r6_bench.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
tail -c 3000 gpurun_out/r6_profiles/r6_bench_headline.json
