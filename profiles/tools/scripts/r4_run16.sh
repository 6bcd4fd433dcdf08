cd /tmp && export TMPDIR=/tmp
CV_ONE=native_K1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_dbg_native -o cv -- python $GRAFT_REPO_ROOT/profiles/tools/cv_bwd_form_ab.py 2>&1 | tail -2
python - <<PY
import csv
rows=list(csv.DictReader(open('$GRAFT_REPO_ROOT/gpurun_out/prof_dbg_native/cv_kernel_stats.csv')))
for r in rows[:12]:
    print(f"{r['Name'][:70]:70s} {r['Calls']:>4s} {float(r['AverageNs'])/1e3:9.1f} {float(r['MaxNs'])/1e3:9.1f}")
PY
