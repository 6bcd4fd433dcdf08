import sys, os, json, subprocess
# A/B of builds of the library on the PTF fold (2 views @ 384x512), one GPU session:
#   AB_LIBS="base=freesplat_amd/lib_base.so,new=" python profiles/tools/ptf_ab.py
variants = [v.split("=", 1) for v in os.environ.get("AB_LIBS", "base=freesplat_amd/lib_base.so,new=").split(",")]
code = ("import bench_encoder as b, torch, json; r = b.bench_ptf(torch.device('cuda:0'), 20, 3, cpu=False); "
        "print(json.dumps({'fold_ms': r['ms_per_call'], 'kernel_ms': r['roofline']['kernel_ms_per_fold'], "
        "'train_ms': r['train_fwd_bwd']['hip_ms']}))")
for tag, lib in variants * 2:
    env = dict(os.environ)
    if lib:
        env["FREESPLAT_LIB"] = os.path.join(os.getcwd(), lib)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    print(tag, lines[-1] if lines else out.stderr[-400:], flush=True)
