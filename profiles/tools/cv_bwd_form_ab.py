"""Same-session A/B of the three forms of the cost-volume training step (saved: training forward keeps the MLP inputs +
two-pass backward; two_pass: the backward recomputes the forward, FREESPLAT_CV_SAVE=0; atomic: the one-kernel atomic
scatter, FS_CV_BWD_ATOMIC=1) on the three benchmarked shapes; one subprocess per form, twice, interleaved.
   python profiles/tools/cv_bwd_form_ab.py [workload ...]        (native_K1 c3scale_K2 fvt10_K8)
With CV_ONE=<workload> it runs ONE training step loop of that workload in-process (for rocprofv3)."""
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
    out[name] = (round(r["roofline"]["avg_launch_ms"], 4), round(r["train_fwd_bwd"]["ms"], 3))
print("RESULT " + json.dumps(out))
'''

if __name__ == "__main__":
    names = [a for a in sys.argv[1:]] or list(WL)
    env0 = dict(os.environ, CV_WL=json.dumps({n: WL[n] for n in names}))
    if os.environ.get("CV_ONE"):
        sys.path.insert(0, ROOT)
        os.chdir(ROOT)
        import torch
        import bench_encoder as b
        kw = dict(WL[os.environ["CV_ONE"]])
        kw.pop("steps"); kw.pop("warmup")
        print(b.bench_cost_volume(torch.device("cuda:0"), 4, 1, cpu=False, **kw)["train_fwd_bwd"])
        sys.exit(0)
    for tag, atomic, save in (("default", "0", ""), ("saved", "0", "1"), ("two_pass", "0", "0"), ("atomic", "1", "0")) * 2:
        p = subprocess.run([sys.executable, "-c", CODE], env=dict(env0, FS_CV_BWD_ATOMIC=atomic, FREESPLAT_CV_SAVE=save),
                           capture_output=True, text=True, cwd=ROOT)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
        print(tag, line[-1][7:] if line else p.stderr[-800:], "(forward ms per call, fwd+bwd ms)", flush=True)
