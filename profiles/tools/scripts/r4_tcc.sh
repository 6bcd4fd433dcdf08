cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for W in cv_c3scale_K2 cv_fvt10_K8 cvt_fvt10_K8; do
  OUT=gpurun_out/tcc_r4/$W
  mkdir -p $OUT
  timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d $OUT -o x --output-format csv -- python profiles/tools/fwd_traffic.py run $W 2 > $OUT/log.txt 2>&1
  python - "$OUT" "$W" <<'PY'
import sys, glob, csv, collections
out, w = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "fs::" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    if m.get("TCC_REQ_sum", 0) > 1e5:
        hit, miss = m.get("TCC_HIT_sum", 0), m.get("TCC_MISS_sum", 0)
        print(f"{w} {k}: TCC_REQ {m.get('TCC_REQ_sum', 0):.3e} HIT {hit:.3e} MISS {miss:.3e} hit rate {hit / max(hit + miss, 1):.3f} per launch ({len(next(iter(d.values())))} launches)")
PY
done 2>&1 | tee gpurun_out/r4_tcc_hit_miss.txt
