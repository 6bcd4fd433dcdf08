#!/usr/bin/env python
"""Where the GPU time of one composed config-3 training step goes, from a rocprofv3 kernel trace of
`python bench_c3_step.py --trace-steps N`: every kernel between the two marker launches
(torch.erfinv on a 64-element tensor before and after the N steps) is put in one of three classes BY NAME --
  library   fs::*                          the hot path's own kernels
  stand-in  MIOpen / rocBLAS / hipBLASLt convolution and GEMM kernels, tanh, bilinear upsampling, ReLU (threshold), concatenation:
            the out-of-scope modules' stand-ins (bench_c3_step.py) -- nothing on the hot path uses these operators
  glue      everything else: the torch elementwise / copy / fill / reduce / index kernels and rocclr copies BETWEEN the hot-path
            stages (encoder_forward's reshapes, sigmoid, the skip add, the loss, gradient accumulation)
and summed per step.  glue_frac_of_hotpath_gpu_time = glue / (library + glue).
Usage: python profiles/tools/c3_step_glue.py <trace dir> <steps> > profiles/r5_c3_step_glue.json"""
import collections
import csv
import glob
import json
import os
import re
import sys

STANDIN = re.compile(r"miopen|MIOpen|igemm|Igemm|naive_conv|gridwise|Cijk_|gemm|Gemm|conv|Conv|tanh|upsample_bilinear|threshold|"
                     r"CatArrayBatchedCopy|col2im|im2col|SubTensorOpWithScalar|transpose_|batched_transpose|wrw|fwd_|bwd_|clamp_min|relu", re.I)


def main(src, steps):
    rows = []
    for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "erfinv" in r["Kernel_Name"]]
    if len(marks) >= 2:
        rows = rows[marks[-2] + 1:marks[-1]]
    cls_ms = collections.Counter()
    by_kernel = collections.defaultdict(lambda: [0.0, 0])
    for r in rows:
        name = r["Kernel_Name"]
        short = name.split("(")[0].replace("void ", "")
        ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
        c = "library" if "fs::" in short else ("standin" if STANDIN.search(name) else "glue")
        cls_ms[c] += ms
        e = by_kernel[(c, short[:110])]
        e[0] += ms
        e[1] += 1
    span_ms = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) * 1e-6 if rows else 0.0
    per = {k: v / steps for k, v in cls_ms.items()}
    top = lambda c, n: [{"kernel": k[1], "ms_per_step": round(v[0] / steps, 4), "launches_per_step": round(v[1] / steps, 1)}
                        for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1][0]) if k[0] == c][:n]
    out = {"what": __doc__.split("\n\n")[0] if "\n\n" in __doc__ else __doc__, "steps": steps, "kernels_in_window": len(rows),
           "per_step_ms": {k: round(per.get(k, 0.0), 4) for k in ("library", "standin", "glue")},
           "window_ms_per_step": round(span_ms / steps, 4),
           "gpu_idle_ms_per_step": round(span_ms / steps - sum(per.values()), 4),
           "glue_frac_of_hotpath_gpu_time": round(per.get("glue", 0.0) / max(per.get("library", 0.0) + per.get("glue", 0.0), 1e-9), 4),
           "launches_per_step": {c: round(sum(v[1] for k, v in by_kernel.items() if k[0] == c) / steps, 1) for c in ("library", "standin", "glue")},
           "top_glue_kernels": top("glue", 25), "top_library_kernels": top("library", 25), "top_standin_kernels": top("standin", 12)}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
