cd $GRAFT_REPO_ROOT
for K in 3 4; do for shape in 1 4 1 4; do
FS_CV_FWD_SHAPE=$shape python - <<PY
import os, torch, bench_encoder as be
r = be.bench_cost_volume(torch.device("cuda:0"), 30, 5, V=5, K=$K, h4=96, w4=128)
print("K=$K shape", os.environ["FS_CV_FWD_SHAPE"], "fwd ms", round(r["ms_per_call"], 4), "train ms", round(r["train_fwd_bwd"]["ms"], 4))
PY
done; done
