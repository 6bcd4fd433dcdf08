cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_composed_dropin.py -q -m gpu 2>&1 | grep -E "passed|failed|Error|^E  |^FAILED" | cut -c1-900 | head -40 | tee gpurun_out/r4_composed.log
timeout 900 python -m pytest tests/test_cost_volume_hip.py tests/test_ptf_hip.py tests/test_configs_4_5.py -q -m gpu -k "golden or backward_forms or k2_behind or match_bit_exact or config4_fold or ragged" 2>&1 | grep -E "passed|failed|Error|^E  |^FAILED" | cut -c1-600 | head -30 | tee gpurun_out/r4_tests_b.log
timeout 600 python profiles/tools/cv_bwd_form_ab.py 2>&1 | tee gpurun_out/r4_cv_bwd_form_ab2.txt
bash profiles/tools/fwd_traffic.sh r4 cv_c3scale_K2 cv_fvt10_K8 cvt_native_K1 cvt_c3scale_K2 cvt_fvt10_K8 2>&1 | tail -12
