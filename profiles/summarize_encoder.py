#!/usr/bin/env python
"""Condense gpurun_out/prof_<tag>_{cv,ptf}/ (profiles/run_rocprof_encoder.sh) into committed summaries:
profiles/<tag>_kernel_stats.csv, <tag>_hbm_traffic.json (FETCH_SIZE doubled: gfx950 correction, MI355X_MICROARCH.md "HBM")
and <tag>_sq_counters.json (mean per launch; for the cost volume the MFMA counters and the derived MFMA utilisation)."""
import collections
import csv
import glob
import json
import sys


def find(src, sub, name):
    hits = glob.glob(f"{src}/{sub}/**/*{name}", recursive=True)
    return hits[0] if hits else None


def main(src, tag, cmd):
    f = find(src, "trace", "kernel_stats.csv")
    rows = list(csv.DictReader(open(f))) if f else []
    ours = [r for r in rows if "fs::" in r["Name"].split("(")[0]]      # (templated kernels read "void fs::name<...>(")
    total = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
    with open(f"profiles/{tag}_kernel_stats.csv", "w") as o:
        o.write("# rocprofv3 --kernel-trace --stats -- " + cmd + "\n")
        o.write("kernel,calls,avg_us,min_us,max_us,pct_of_gpu_time\n")
        for r in ours:
            o.write(f"{r['Name'].split('(')[0].replace('void ', '')},{r['Calls']},{float(r['AverageNs'])/1e3:.1f},"
                    f"{float(r['MinNs'])/1e3:.1f},{float(r['MaxNs'])/1e3:.1f},{100*float(r['TotalDurationNs'])/total:.2f}\n")
        other = total - sum(float(r["TotalDurationNs"]) for r in ours)
        o.write(f"(torch / rocBLAS / rocclr kernels of the host glue),,,,,{100*other/total:.2f}\n")

    def mean_counters(sub):
        f = find(src, sub, "counter_collection.csv")
        agg = collections.defaultdict(list)
        if f:
            for r in csv.DictReader(open(f)):
                if "fs::" in r["Kernel_Name"].split("(")[0]:
                    agg[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
        return {k: sum(v) / len(v) for k, v in agg.items()}

    traffic = collections.defaultdict(dict)
    for sub, key in (("fetch", "fetch_size_kb_raw"), ("write", "write_size_kb_raw")):
        for (k, _), v in mean_counters(sub).items():
            traffic[k][key] = v
    tr = {}
    for k, d in traffic.items():
        rd, wr = 2.0 * d.get("fetch_size_kb_raw", 0.0) * 1024, d.get("write_size_kb_raw", 0.0) * 1024
        tr[k] = dict(d, read_bytes=rd, write_bytes=wr, hbm_bytes_per_launch=rd + wr)
    json.dump({"command": cmd, "note": "read_bytes = 2 * FETCH_SIZE * 1024 (gfx950 correction), write_bytes = WRITE_SIZE * 1024; "
               "separate --pmc passes; mean per launch", "kernels": tr}, open(f"profiles/{tag}_hbm_traffic.json", "w"), indent=1)
    sq = collections.defaultdict(dict)
    for sub in ("sq1", "sq2"):
        for (k, c), v in mean_counters(sub).items():
            sq[k][c] = v
    avg_us = {r["Name"].split("(")[0].replace("void ", ""): float(r["AverageNs"]) / 1e3 for r in ours}
    for k, d in sq.items():
        if d.get("SQ_VALU_MFMA_BUSY_CYCLES") and avg_us.get(k):
            # SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles summed over the 1024 SIMDs (MI355X_MICROARCH.md, PMC units);
            # utilisation = busy cycles / (SIMDs x kernel duration x 2.4 GHz); duration from the kernel-trace pass
            d["avg_us_kernel_trace"] = avg_us[k]
            d["mfma_busy_frac_at_2.4GHz"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * avg_us[k] * 2400.0)
    json.dump({"command": cmd, "note": "mean per launch, separate --pmc passes", "kernels": sq},
              open(f"profiles/{tag}_sq_counters.json", "w"), indent=1)
    print(open(f"profiles/{tag}_kernel_stats.csv").read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
