cd $GRAFT_REPO_ROOT
for sv in 1 0 1 0; do
FREESPLAT_GRU_SAVE=$sv python - <<'PY'
import json, os, torch, bench_encoder as be
dev = torch.device("cuda:0")
r = be.bench_ptf(dev, 40, 5)
t = r["train_fwd_bwd"]
print("SAVE", os.environ["FREESPLAT_GRU_SAVE"], "fold2 infer ms", round(r["ms_per_call"], 4), "train hip_ms", round(t["hip_ms"], 4), "kernel_ms", round(t["roofline"]["kernel_ms_per_step"], 4))
PY
done
