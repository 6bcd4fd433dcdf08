"""Same-session A/B of builds of the library on the cost volume's training step (and forward call), one subprocess per build,
twice, interleaved:   AB_LIBS="base=,th4=freesplat_amd/libfreesplat_hip_sgth4.so" python profiles/tools/cv_lib_ab.py [workload ...]
(workloads: native_K1 c3scale_K2 fvt10_K8; an empty path = the in-tree library)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
WL = {"native_K1": dict(steps=20, warmup=3), "c3scale_K2": dict(steps=6, warmup=2, V=3, K=2, h4=242, w4=324),
      "fvt10_K8": dict(steps=6, warmup=2, V=10, K=8)}
CODE = r'''
import sys, os, json; sys.path.insert(0, os.getcwd())
import bench_encoder as b, torch
d = torch.device("cuda:0")
WL = json.loads(os.environ["CV_WL"])
out = {}
for name, kw in WL.items():
    st, wu = kw.pop("steps"), kw.pop("warmup")
    r = b.bench_cost_volume(d, st, wu, cpu=False, **kw)
    out[name] = {"fwd_ms": round(r["ms_per_call"], 4), "train_ms": round(r["train_fwd_bwd"]["ms"], 3)}
print("RESULT " + json.dumps(out))
'''
names = sys.argv[1:] or list(WL)
variants = [v.split("=", 1) for v in os.environ.get("AB_LIBS", "base=").split(",")]
for tag, lib in variants * 2:
    env = dict(os.environ, CV_WL=json.dumps({n: WL[n] for n in names}))
    if lib:
        env["FREESPLAT_LIB"] = os.path.join(ROOT, lib)
    r = subprocess.run([sys.executable, "-c", CODE], cwd=ROOT, env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    print(tag, line[-1][7:] if line else "FAILED " + r.stderr[-800:], flush=True)
