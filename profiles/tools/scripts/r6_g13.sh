cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/g13_tests.log 2>&1
tail -5 gpurun_out/g13_tests.log
timeout 900 python bench.py > gpurun_out/g13_bench.log 2>&1
tail -c 4000 gpurun_out/g13_bench.log
