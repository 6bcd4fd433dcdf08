cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ptf_hip.py tests/test_composed_dropin.py tests/test_configs_4_5.py -q -m gpu 2>&1 | grep -E "passed|failed|^E  |^FAILED" | cut -c1-500 | head -8
code="import bench_encoder as b, torch, json; r = b.bench_ptf(torch.device('cuda:0'), 20, 3, cpu=False); r10 = b.bench_ptf(torch.device('cuda:0'), 5, 1, V=10, cpu=False, train=False); print(json.dumps({'fold_ms': r['ms_per_call'], 'kernel_ms': r['roofline']['kernel_ms_per_fold'], 'train_ms': r['train_fwd_bwd']['hip_ms'], 'fold10_ms': r10['ms_per_call']}))"
for rep in 1 2; do
  for v in 1 0; do
    echo -n "FS_PTF_SERIAL=$v: "; FS_PTF_SERIAL=$v python -c "$code" 2>&1 | grep "^{" | tail -1
  done
done 2>&1 | tee gpurun_out/r4_ptf_side_stream_ab.txt
