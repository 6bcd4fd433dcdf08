"""Per-step wall times (synchronised) of the native cost-volume training step, forward and backward separately, with the
module's one-launch depth planes and with the torch formulation of generate_depth_planes (two depth values -> the fallback path):
how round 4 found the one-time 60 ms in the SECOND backward (torch loading its `add` kernel for the first .grad accumulation once the
forward no longer launched torch kernels), which a five-step timed region with one warm-up step reported as a 3x regression.
   python profiles/tools/cv_step_times.py"""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden"))
import torch, inputs
from freesplat_amd.cost_volume import AVGFeatureVolumeManager
dev = torch.device("cuda:0")
V, K, h4, w4, D, C = 2, 1, 96, 128, 128, 48
torch.manual_seed(0)
m = AVGFeatureVolumeManager(h4, w4, num_depth_bins=D, mlp_channels=[202, 32, 32, 1], matching_dim_size=C).to(dev)
a = {k: v.to(dev) for k, v in inputs.cv_inputs(V, K, h4, w4, C, seed=1).items()}
a["cur_feats"].requires_grad_(True); a["src_feats"].requires_grad_(True)
for mode in ("kernel_planes", "torch_planes"):
    if mode == "torch_planes":
        a["min_depth"] = a["min_depth"].expand(2, 1, 1, 1).contiguous()   # numel 2 -> the module's generate_depth_planes path
        a["max_depth"] = a["max_depth"].expand(2, 1, 1, 1).contiguous()
    ts = []
    for i in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        o = m(**a); torch.cuda.synchronize(); t1 = time.perf_counter()
        o.backward(torch.ones_like(o)); torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append((round((t1 - t0) * 1e3, 2), round((t2 - t1) * 1e3, 2)))
    print(mode, ts)
