cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r5_bench_stdout.log 2> gpurun_out/r5_bench_stderr.log
tail -c 300 gpurun_out/r5_bench_stderr.log
tail -n 1 gpurun_out/r5_bench_stdout.log | wc -c
tail -n 1 gpurun_out/r5_bench_stdout.log | tail -c 700
