cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 ) > gpurun_out/r5_gpu_tests.log; tail -4 gpurun_out/r5_gpu_tests.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r5_bench_stdout.log 2> gpurun_out/r5_bench_stderr.log
tail -c 400 gpurun_out/r5_bench_stderr.log
tail -n 1 gpurun_out/r5_bench_stdout.log | head -c 4200; echo
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
