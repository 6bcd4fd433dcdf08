import sys, os; sys.path.insert(0, os.getcwd())
import bench_encoder as b, torch
d = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "native"
if which == "native":
    r = b.bench_cost_volume(d, 8, 2, cpu=False)
elif which == "c3":
    r = b.bench_cost_volume(d, 4, 1, V=3, K=2, h4=242, w4=324, cpu=False)
else:
    r = b.bench_cost_volume(d, 4, 1, V=10, K=8, cpu=False)
print(which, "fwd ms", round(r["ms_per_call"], 3))
