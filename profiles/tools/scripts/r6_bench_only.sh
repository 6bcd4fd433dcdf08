cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6_profiles
export TMPDIR=/tmp
timeout 1500 python bench.py > gpurun_out/r6_profiles/r6_bench_stdout.log 2>&1
tail -1 gpurun_out/r6_profiles/r6_bench_stdout.log > gpurun_out/r6_profiles/r6_bench_headline.json
cp gpurun_out/bench_full.json gpurun_out/r6_profiles/r6_bench.json
tail -c 1200 gpurun_out/r6_profiles/r6_bench_headline.json
