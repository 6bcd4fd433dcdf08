cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -8 > gpurun_out/g10_tests.log
cat gpurun_out/g10_tests.log
timeout 120 rocprofv3 -L 2>/dev/null | grep -iE "^\s*(gpu|Name|.*(TCP_|TA_|TD_)[A-Z_]+)" | grep -oE "(TCP|TA|TD)_[A-Za-z_0-9]+" | sort -u | tr '\n' ' ' > gpurun_out/g10_counters.txt
wc -c gpurun_out/g10_counters.txt
CMD="python profiles/tools/cv_prof.py fvt10"
i=0
for S in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum"; do
  timeout 300 rocprofv3 --pmc $S --kernel-trace -d gpurun_out/tcp_p$i -o x --output-format csv -- $CMD > gpurun_out/g10_tcp_p$i.log 2>&1
  i=$((i+1))
done
python - <<'PY' > gpurun_out/g10_tcp_summary.txt
import glob, csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/tcp_p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "cost_volume" in k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/tcp_p0/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "cost_volume" in k: dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, d in agg.items():
    print(k, "avg_us", sum(dur[k]) / max(len(dur[k]), 1))
    for c, v in d.items(): print(f"   {c:44s} {sum(v)/len(v):18.1f}   ({len(v)} launches)")
PY
cat gpurun_out/g10_tcp_summary.txt; tail -3 gpurun_out/g10_tcp_p*.log | grep -iE "error|abort|exceed" | head
rm -rf gpurun_out/tcp_p*
python bench.py > gpurun_out/g10_bench.log 2>&1; tail -1 gpurun_out/g10_bench.log | cut -c1-1500
