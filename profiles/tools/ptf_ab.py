import sys, os; sys.path.insert(0, os.getcwd())
import bench_encoder as b, torch
d = torch.device("cuda:0")
for V in (2, 10):
    r = b.bench_ptf(d, 10 if V == 2 else 4, 2, V=V, cpu=False)
    print(f"V={V}: {r['ms_per_call']:.3f} ms/call, kernels {r['roofline']['kernel_ms_per_fold']:.3f} ms, frac {r['roofline']['frac']:.3f}, train {r['train_fwd_bwd']['hip_ms']:.2f} ms")
