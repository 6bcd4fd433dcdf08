cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_cost_volume_hip.py tests/test_composed_dropin.py tests/test_pipeline_c1.py -q -m gpu -k "not (backward_tight and native)" 2>&1 | grep -E "passed|failed|^E  |^FAILED" | cut -c1-600 | head -10
timeout 600 python profiles/tools/cv_bwd_form_ab.py native_K1 c3scale_K2 2>&1 | grep -v saved | tee gpurun_out/r4_cv_prep_fused_ab.txt
python -c "
import sys; sys.path.insert(0,'.')
import torch, bench_encoder as b
r=b.bench_cost_volume(torch.device('cuda:0'), 40, 5, cpu=False)
print('native forward ms_per_call', round(r['ms_per_call'],4), 'train', round(r['train_fwd_bwd']['ms'],4))"
bash profiles/tools/fwd_traffic.sh r4 cvt_native_K1 cvt_c3scale_K2 cvt_fvt10_K8 cv_native_K1 2>&1 | tail -5
