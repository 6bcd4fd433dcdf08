import sys, os, json, subprocess
# A/B of two builds of the library on the same GPU box: python profiles/tools/raster_ab.py [train]
mode = ["--mode", "train", "--views", "8", "--steps", "10"] if len(sys.argv) > 1 and sys.argv[1] == "train" else []
for tag, lib in (("base", os.path.join(os.getcwd(), "freesplat_amd", "lib_base.so")), ("new", ""), ("base", os.path.join(os.getcwd(), "freesplat_amd", "lib_base.so")), ("new", "")):
    env = dict(os.environ)
    if lib:
        env["FREESPLAT_LIB"] = lib
    out = subprocess.run([sys.executable, "bench.py", "--sections", "raster", "--no-cpu-baseline", "--no-graph"] + mode, env=env, capture_output=True, text=True).stdout
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    print(tag, round(d["value"], 1), {k: round(v, 4) for k, v in d["kernel_ms_per_view"].items()}, flush=True)
