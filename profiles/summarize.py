#!/usr/bin/env python
"""Condense rocprofv3 output directories (gpurun_out/prof_*/{trace,fetch,write}) into the small,
committed summaries under profiles/: per-kernel time stats of the library's kernels and per-launch
HBM traffic from the FETCH_SIZE / WRITE_SIZE passes (gfx950 correction: FETCH_SIZE counts 64 B per
128-B request for wide coalesced reads -> doubled, MI355X_MICROARCH.md "HBM")."""
import collections
import csv
import json
import sys


def main(src, tag, cmd):
    rows = list(csv.DictReader(open(f"{src}/trace/c3_kernel_stats.csv")))
    ours = [r for r in rows if "fs::" in r["Name"].split("(")[0]]   # (templated kernels read "void fs::name<...>(")
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(f"profiles/{tag}_kernel_stats.csv", "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- " + cmd + "\n")
        f.write("kernel,calls,avg_us,min_us,max_us,pct_of_gpu_time\n")
        for r in ours:
            f.write(f"{r['Name'].split('(')[0].replace('void ', '')},{r['Calls']},{float(r['AverageNs'])/1e3:.1f},"
                    f"{float(r['MinNs'])/1e3:.1f},{float(r['MaxNs'])/1e3:.1f},{100*float(r['TotalDurationNs'])/total:.2f}\n")
        other = total - sum(float(r["TotalDurationNs"]) for r in ours)
        f.write(f"(torch/rocclr kernels of the host framing),,,,,{100*other/total:.2f}\n")
    traffic = collections.defaultdict(dict)
    for name in ("fetch", "write"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f"{src}/{name}/c3_counter_collection.csv")):
            if "fs::" in r["Kernel_Name"].split("(")[0]:
                agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            traffic[k][name + "_size_kb_raw"] = sum(v) / len(v)
    out = {}
    for k, d in traffic.items():
        rd = 2.0 * d.get("fetch_size_kb_raw", 0.0) * 1024
        wr = d.get("write_size_kb_raw", 0.0) * 1024
        out[k] = dict(d, read_bytes=rd, write_bytes=wr, hbm_bytes_per_launch=rd + wr)
    json.dump({"command": cmd, "note": "read_bytes = 2 * FETCH_SIZE * 1024 (gfx950 correction), "
               "write_bytes = WRITE_SIZE * 1024; separate --pmc passes", "kernels": out},
              open(f"profiles/{tag}_hbm_traffic.json", "w"), indent=1)
    sq = sq_summary(src)
    if sq:
        json.dump(sq, open(f"profiles/{tag}_sq_counters.json", "w"), indent=1)
    print(open(f"profiles/{tag}_kernel_stats.csv").read())
    print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in out.items()}), "MB per launch")
    if sq:
        print(json.dumps({k: {c: round(v.get(c, 0) / 1e6, 2) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS")}
                          for k, v in sq.items()}), "M wave-instructions per launch")


def sq_summary(src):
    """Mean per-launch SQ counters of the library's kernels from the sq1/sq2 passes."""
    out = collections.defaultdict(dict)
    for name in ("sq1", "sq2"):
        try:
            rows = csv.DictReader(open(f"{src}/{name}/c3_counter_collection.csv"))
        except FileNotFoundError:
            continue
        agg = collections.defaultdict(list)
        for r in rows:
            if "fs::" in r["Kernel_Name"].split("(")[0]:
                agg[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            out[k][c] = sum(v) / len(v)
    return out


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
