import sys, os; sys.path.insert(0, os.getcwd())
import bench_encoder as b, torch
d = torch.device("cuda:0")
r = b.bench_cost_volume(d, 8, 2, cpu=False); print("native fwd+bwd ms", round(r["train_fwd_bwd"]["ms"], 3), "fwd ms", round(r["ms_per_call"], 3))
r = b.bench_cost_volume(d, 4, 1, V=3, K=2, h4=242, w4=324, cpu=False); print("c3 fwd+bwd ms", round(r["train_fwd_bwd"]["ms"], 2), "fwd ms", round(r["ms_per_call"], 3))
