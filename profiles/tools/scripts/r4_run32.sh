cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_adapter_hip.py tests/test_depth_tail.py tests/test_composed_dropin.py tests/test_pipeline_c1.py -q -m gpu 2>&1 | grep -E "passed|failed|^E  |^FAILED" | cut -c1-700 | head -14
export TMPDIR=/tmp
( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r4_adapter3 -o x -- python $GRAFT_REPO_ROOT/profiles/tools/adapter_prof.py 2>&1 | grep -E "ms/step|Error|error" | head -5 )
f=$(find gpurun_out/prof_r4_adapter3 -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:10]:
    print(r['Name'][:90].ljust(90), r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
