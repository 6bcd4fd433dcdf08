"""Summarise a rocprofv3 --kernel-trace CSV: per kernel calls, avg / min / max us, share.  python profiles/tools/kstats.py <dir> [header]"""
import sys, glob, csv, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in d.values())
if len(sys.argv) > 2: print("# " + sys.argv[2])
print("kernel,calls,avg_us,min_us,max_us,pct_of_gpu_time")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[:90]},{len(v)},{sum(v)/len(v):.1f},{min(v):.1f},{max(v):.1f},{100*sum(v)/tot:.2f}")
