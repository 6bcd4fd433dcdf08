cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_composed_dropin.py -q -m gpu 2>&1 | grep -E "passed|failed|Error|^E  |^FAILED" | cut -c1-2500 | head -30 | tee gpurun_out/r4_tests_c.log
