cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
FREESPLAT_LIB=$PWD/freesplat_amd/libfreesplat_hip_quadproj.so FS_CV_PROJECTED=0 timeout 600 python -m pytest tests/test_cost_volume_hip.py tests/test_configs_4_5.py -q -m gpu -k "golden or native_size or ragged or config4_cost_volume or exact_zero or oblique" 2>&1 | grep -E "passed|failed|^E  |^FAILED" | cut -c1-400 | head -8
cat > /tmp/fw.py <<'PY'
import sys, os, json
sys.path.insert(0, os.getcwd())
import torch, bench_encoder as b
d = torch.device("cuda:0")
out = {}
for name, kw in (("c3scale_K2", dict(steps=12, warmup=3, V=3, K=2, h4=242, w4=324)), ("fvt5_K4", dict(steps=12, warmup=3, V=5, K=4)),
                 ("fvt10_K8", dict(steps=12, warmup=3, V=10, K=8))):
    st, wu = kw.pop("steps"), kw.pop("warmup")
    r = b.bench_cost_volume(d, st, wu, cpu=False, **kw)
    out[name] = round(r["roofline"]["avg_launch_ms"], 4)
print("RESULT " + json.dumps(out))
PY
for i in 1 2; do for v in quadproj ""; do
  if [ -z "$v" ]; then lib=$PWD/freesplat_amd/libfreesplat_hip.so; else lib=$PWD/freesplat_amd/libfreesplat_hip_$v.so; fi
  echo -n "${v:-base} "; FREESPLAT_LIB=$lib python /tmp/fw.py 2>&1 | grep RESULT
done; done | tee gpurun_out/r4_cv_quadproj_ab.txt
