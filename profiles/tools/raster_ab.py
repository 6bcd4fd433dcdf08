import sys, os, json, subprocess
# A/B of builds of the library on the same GPU box (box-to-box variation is +-5 %):
#   AB_LIBS="base=freesplat_amd/lib_base.so,new=" python profiles/tools/raster_ab.py [train]
# an empty path = the in-tree libfreesplat_hip.so; every variant is run twice, interleaved.
mode = ["--mode", "train", "--views", "8", "--steps", "10"] if len(sys.argv) > 1 and sys.argv[1] == "train" else []
mode += os.environ.get("AB_ARGS", "").split()      # e.g. AB_ARGS="--workload c3_closeup_968x1296_1M --views 4 --steps 5"
variants = [v.split("=", 1) for v in os.environ.get("AB_LIBS", "base=freesplat_amd/lib_base.so,new=").split(",")]
for tag, lib in variants * 2:
    env = dict(os.environ)
    if lib:
        env["FREESPLAT_LIB"] = os.path.join(os.getcwd(), lib)
    res = subprocess.run([sys.executable, "bench.py", "--sections", "raster", "--no-cpu-baseline", "--no-graph"] + mode, env=env, capture_output=True, text=True)
    out = res.stdout
    if res.returncode != 0 or not [l for l in out.splitlines() if l.startswith("{")]:
        print(tag, "FAILED rc", res.returncode, res.stderr[-1500:], flush=True)
        continue
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][-2])   # the full line (the compact one follows it)
    print(tag, round(d["value"], 1), {k: round(v, 4) for k, v in d["kernel_ms_per_view"].items()}, flush=True)
