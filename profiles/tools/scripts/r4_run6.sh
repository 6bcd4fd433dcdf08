cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_composed_dropin.py tests/test_ptf_hip.py tests/test_raster_hip.py -q -m gpu -k "composed or invert_4x4 or overflow or capacity" 2>&1 | grep -E "passed|failed|Error|^E  |^FAILED" | cut -c1-1200 | head -30 | tee gpurun_out/r4_tests_c.log
( time timeout 900 python bench.py > gpurun_out/r4_bench_try.json 2> gpurun_out/r4_bench_try.err ) 2>&1 | tail -3
tail -c 600 gpurun_out/r4_bench_try.err
python profiles/tools/bench_digest.py gpurun_out/r4_bench_try.json
