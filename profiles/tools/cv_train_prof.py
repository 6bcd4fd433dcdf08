"""A few cost-volume TRAINING steps (forward + backward) of one benchmark shape, for rocprofv3 --kernel-trace --stats:
   python profiles/tools/cv_train_prof.py native | c3 | fvt10 [steps]"""
import sys, os; sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden"))
import time, torch, inputs
from freesplat_amd.cost_volume import AVGFeatureVolumeManager
which = sys.argv[1] if len(sys.argv) > 1 else "native"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
V, K, h4, w4 = {"native": (2, 1, 96, 128), "c3": (3, 2, 242, 324), "fvt10": (10, 8, 96, 128)}[which]
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = AVGFeatureVolumeManager(matching_height=h4, matching_width=w4, num_depth_bins=128, mlp_channels=[202, 32, 32, 1],
                            matching_dim_size=48).to(dev)
a = {k: v.to(dev) for k, v in inputs.cv_inputs(V, K, h4, w4, 48, seed=1).items()}
a["cur_feats"].requires_grad_(True); a["src_feats"].requires_grad_(True)
cot = torch.ones(V, 128, h4, w4, device=dev)
leaves = [a["cur_feats"], a["src_feats"]] + list(m.parameters())
def step():
    for t in leaves: t.grad = None          # (no AccumulateGrad `add` kernels: gradients start from None as after zero_grad)
    o = m(**a); o.backward(cot)
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize()
print(which, "train step ms", round((time.perf_counter() - t0) / steps * 1e3, 3))
