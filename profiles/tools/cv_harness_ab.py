import sys, os; sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden"))
import time, torch, inputs
from freesplat_amd.cost_volume import AVGFeatureVolumeManager
dev = torch.device("cuda:0")
for which,(V,K,h4,w4) in {"c3": (3, 2, 242, 324), "fvt10": (10, 8, 96, 128), "native": (2,1,96,128)}.items():
    torch.manual_seed(0)
    m = AVGFeatureVolumeManager(matching_height=h4, matching_width=w4, num_depth_bins=128, mlp_channels=[202, 32, 32, 1], matching_dim_size=48).to(dev)
    a = {k: v.to(dev) for k, v in inputs.cv_inputs(V, K, h4, w4, 48, seed=1).items()}
    a["cur_feats"].requires_grad_(True); a["src_feats"].requires_grad_(True)
    cot = torch.ones(V, 128, h4, w4, device=dev)
    leaves = [a["cur_feats"], a["src_feats"]] + list(m.parameters())
    def old():
        o = m(**a); o.backward(torch.ones_like(o))
    def new():
        for t in leaves: t.grad = None
        o = m(**a); o.backward(cot)
    res = {}
    for rep in range(3):
        for name, fn in (("accumulate", old), ("from_none", new)):
            for _ in range(2): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(8): fn()
            torch.cuda.synchronize()
            res.setdefault(name, []).append(round((time.perf_counter() - t0) / 8 * 1e3, 3))
    print(which, res)
