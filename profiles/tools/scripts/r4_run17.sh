cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_cost_volume_hip.py -q -m gpu -k "depth_planes" 2>&1 | grep -E "passed|failed|^E  |^FAILED" | cut -c1-400 | head -6
timeout 600 python profiles/tools/cv_bwd_form_ab.py native_K1 c3scale_K2 fvt10_K8 2>&1 | grep -v "saved\|atomic" | tee gpurun_out/r4_cv_prep_fused_ab.txt
