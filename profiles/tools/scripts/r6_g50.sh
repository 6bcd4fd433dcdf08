cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_ptf_hip.py tests/test_configs_4_5.py tests/test_composed_dropin.py tests/test_pipeline_c1.py tests/test_compat_reference.py -m gpu -q 2>&1 | tail -4
python bench_c3_step.py --steps 3 --warmup 2 2>&1 | tail -2 | cut -c1-1500
