cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ptf_hip.py tests/test_composed_dropin.py -q -m gpu 2>&1 | grep -E "passed|failed|^E  |^FAILED" | cut -c1-600 | head -10
code="import bench_encoder as b, torch, json; r = b.bench_ptf(torch.device('cuda:0'), 20, 3, cpu=False); print(json.dumps({'fold_ms': r['ms_per_call'], 'train_ms': r['train_fwd_bwd']['hip_ms']}))"
cp freesplat_amd/ptf.py /tmp/ptf_new.py
for rep in 1 2; do
  for v in base new; do
    if [ $v = base ]; then cp _ptf_base.py freesplat_amd/ptf.py; else cp /tmp/ptf_new.py freesplat_amd/ptf.py; fi
    echo -n "$v: "; python -c "$code" 2>&1 | grep "^{" | tail -1
  done
done 2>&1 | tee gpurun_out/r4_ptf_dw_ab.txt
cp /tmp/ptf_new.py freesplat_amd/ptf.py
export TMPDIR=/tmp
for nf in 0; do
  if [ $nf = 1 ]; then export FS_PTF_DW_NOFLUSH=1; fi
  ( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r4_ptfdw$nf -o x -- python -c "import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT'); import os; os.chdir('$GRAFT_REPO_ROOT'); import bench_encoder as b, torch; b.bench_ptf(torch.device('cuda:0'), 5, 2, cpu=False)" > /dev/null 2>&1 )
  f=$(find gpurun_out/prof_r4_ptfdw$nf -name "*kernel_stats.csv" | head -1)
  echo "noflush=$nf"
  python - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(r['Name'][:80].ljust(80), r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
done 2>&1 | tee gpurun_out/r4_ptf_dw_trace.txt
