cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_cost_volume_hip.py -q -m gpu -k "depth_planes" 2>&1 | grep -E "passed|failed|^E  |^FAILED" | cut -c1-400 | head -6
cat > /tmp/par.py <<'PY'
import sys, os, json
sys.path.insert(0, os.getcwd())
import torch, bench_encoder as b
d = torch.device("cuda:0")
out = {}
for name, kw in (("native_K1", dict(steps=20, warmup=3)), ("c3scale_K2", dict(steps=6, warmup=2, V=3, K=2, h4=242, w4=324, cpu_views=1)),
                 ("fvt10_K8", dict(steps=6, warmup=2, V=10, K=8, cpu_views=1))):
    st, wu = kw.pop("steps"), kw.pop("warmup")
    r = b.bench_cost_volume(d, st, wu, cpu=True, **kw)
    out[name] = dict(fwd_ms=round(r["roofline"]["avg_launch_ms"], 4), train_ms=round(r["train_fwd_bwd"]["ms"], 3),
                     max_err=r["parity"]["max_abs_err_vs_oracle"], above=r["parity"]["cells_above_1e-4"])
print("RESULT " + json.dumps(out))
PY
for i in 1 2; do for v in contracton ""; do
  if [ -z "$v" ]; then lib=$PWD/freesplat_amd/libfreesplat_hip.so; else lib=$PWD/freesplat_amd/libfreesplat_hip_$v.so; fi
  echo -n "${v:-fast} "; FREESPLAT_LIB=$lib python /tmp/par.py 2>&1 | grep RESULT
done; done | tee gpurun_out/r4_cv_contract_ab.txt
