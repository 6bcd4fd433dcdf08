"""A/B of library builds on the cost-volume workloads inside one GPU session:
   AB_LIBS="lin=freesplat_amd/libfreesplat_hip_cvlin.so,new=" python profiles/tools/cv_ab.py
(an empty path = the in-tree libfreesplat_hip.so; one subprocess per variant, twice, interleaved)."""
import json
import os
import subprocess
import sys

CODE = r'''
import sys, os, json; sys.path.insert(0, os.getcwd())
import bench_encoder as b, torch
d = torch.device("cuda:0")
out = {}
for name, kw in (("native_K1", dict(steps=20, warmup=3)), ("c3scale_K2", dict(steps=5, warmup=2, V=3, K=2, h4=242, w4=324)),
                 ("fvt10_K8", dict(steps=5, warmup=2, V=10, K=8))):
    st, wu = kw.pop("steps"), kw.pop("warmup")
    r = b.bench_cost_volume(d, st, wu, cpu=False, **kw)
    out[name] = (round(r["roofline"]["frac"], 4), round(r["roofline"]["avg_launch_ms"], 4), round(r["train_fwd_bwd"]["ms"], 3))
print("RESULT " + json.dumps(out))
'''
variants = [v.split("=", 1) for v in os.environ.get("AB_LIBS", "base=freesplat_amd/lib_base.so,new=").split(",")]
for tag, lib in variants * 2:
    env = dict(os.environ)
    if lib:
        env["FREESPLAT_LIB"] = os.path.join(os.getcwd(), lib)
    p = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
    print(tag, line[-1][7:] if line else p.stderr[-800:], "(frac of fp32-MFMA peak, forward ms per call, fwd+bwd ms)", flush=True)
