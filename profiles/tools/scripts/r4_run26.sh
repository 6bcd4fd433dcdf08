cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ptf_hip.py tests/test_composed_dropin.py tests/test_configs_4_5.py -q -m gpu 2>&1 | grep -E "passed|failed|^E  |^FAILED" | cut -c1-600 | head -10
python profiles/tools/ptf_call_breakdown.py 2>&1 | tail -2 | tee gpurun_out/r4_ptf_call_breakdown_after.txt
code="import bench_encoder as b, torch, json; r = b.bench_ptf(torch.device('cuda:0'), 20, 3, cpu=False); print(json.dumps({'fold_ms': r['ms_per_call'], 'train_ms': r['train_fwd_bwd']['hip_ms']}))"
python -c "$code" 2>&1 | grep "^{" | tail -1
