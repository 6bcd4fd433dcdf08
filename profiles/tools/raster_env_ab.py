import sys, os, json, subprocess
# Same-box A/B of library builds AND environment switches (box-to-box variation is +-5 %):
#   AB_VARIANTS="legacy:FREESPLAT_PREPROCESS=legacy|b8:FREESPLAT_RASTER_BATCH=8|pw5:FREESPLAT_LIB=freesplat_amd/libfreesplat_hip_pw5.so" \
#       python profiles/tools/raster_env_ab.py [train]
# every variant is run AB_REPEAT (default 2) times, interleaved; AB_ARGS = extra bench.py arguments.
mode = ["--mode", "train", "--views", "8", "--steps", "10"] if len(sys.argv) > 1 and sys.argv[1] == "train" else []
mode += os.environ.get("AB_ARGS", "").split()
variants = []
for item in os.environ.get("AB_VARIANTS", "base:").split("|"):
    tag, _, envs = item.partition(":")
    variants.append((tag, dict(e.split("=", 1) for e in envs.split(";") if e)))
for tag, envs in variants * int(os.environ.get("AB_REPEAT", "2")):
    env = dict(os.environ)
    for k, v in envs.items():
        env[k] = os.path.join(os.getcwd(), v) if k == "FREESPLAT_LIB" else v
    res = subprocess.run([sys.executable, "bench.py", "--sections", "raster", "--no-cpu-baseline", "--no-graph"] + mode, env=env, capture_output=True, text=True)
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    if res.returncode != 0 or not lines:
        print(tag, "FAILED rc", res.returncode, res.stderr[-1500:], flush=True)
        continue
    d = json.loads(lines[-2] if len(lines) > 1 else lines[-1])   # the full line (the compact one follows it)
    print(tag, round(d["value"], 1), {k: round(v, 4) for k, v in d.get("kernel_ms_per_view", {}).items()}, flush=True)
