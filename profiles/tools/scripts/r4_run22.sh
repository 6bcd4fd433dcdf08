cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ptf_hip.py tests/test_composed_dropin.py tests/test_configs_4_5.py -q -m gpu 2>&1 | grep -E "passed|failed|^E  |^FAILED" | cut -c1-500 | head -10
code="import bench_encoder as b, torch, json; r = b.bench_ptf(torch.device('cuda:0'), 20, 3, cpu=False); print(json.dumps({'fold_ms': r['ms_per_call'], 'train_ms': r['train_fwd_bwd']['hip_ms']}))"
cp freesplat_amd/ptf.py /tmp/ptf_new.py
for rep in 1 2; do
  for v in base new; do
    if [ $v = base ]; then cp _ptf_base.py freesplat_amd/ptf.py; else cp /tmp/ptf_new.py freesplat_amd/ptf.py; fi
    echo -n "$v: "; python -c "$code" 2>&1 | grep "^{" | tail -1
  done
done 2>&1 | tee gpurun_out/r4_ptf_glue_ab.txt
cp /tmp/ptf_new.py freesplat_amd/ptf.py
