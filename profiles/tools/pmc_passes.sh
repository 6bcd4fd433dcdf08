#!/bin/bash
# Generic PMC passes of one command, each counter set in its own run (kernel-trace only beside it):
#   profiles/tools/pmc_passes.sh <outdir> <kernel-name-substring> -- <command...>
# prints the mean per-launch value of every counter for the kernels whose name contains the substring.
# (SQ / GRBM sets only: four TA_*_sum counters in one pass made rocprofv3 abort -- "exceeds the capabilities of the
#  hardware" -- and then hang until killed; every pass runs under `timeout`.)
export TMPDIR=/tmp
OUT=$1; PAT=$2; shift 3
mkdir -p $OUT
SETS=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
 "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU"
 "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES"
)
i=0
for S in "${SETS[@]}"; do
  timeout 300 rocprofv3 --pmc $S --kernel-trace -d $OUT/p$i -o x --output-format csv -- "$@" > $OUT/p$i.log 2>&1
  i=$((i+1))
done
python - "$OUT" "$PAT" <<'PY'
import sys, glob, csv, collections
out, pat = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if pat in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in d.items():
        print(f"   {c:44s} {sum(v)/len(v):16.1f}   ({len(v)} launches)")
PY
