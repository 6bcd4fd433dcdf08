"""Exact (contract exp) vs hardware-exp blend on the bench workload, on the GPU, all views of one decoder call:
throughput of both modes in ONE session + how far the images / gradients of the hardware-exp mode are from the exact ones.

    python profiles/tools/exp_mode_ab.py [workload] [views]
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from freesplat_amd import rasterizer as R, synthetic  # noqa: E402
from freesplat_amd.decoder import render_views  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "c3_968x1296_1M"
views = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda", 0)
H, W, N = synthetic.WORKLOADS[workload]
scene = synthetic.make_scene(N)
cams = {k: v.to(dev) for k, v in synthetic.target_cameras(views).items()}
g = {k: scene[k].to(dev).requires_grad_(True) for k in ("means", "covariances", "harmonics", "opacities")}
bg = torch.zeros(views, 3, device=dev)
target = torch.rand(views, 3, H, W, device=dev)


def run(fast: bool, train: bool, steps: int):
    R.FAST_EXP = fast
    def step():
        if train:
            for t in g.values():
                t.grad = None
            c, d = render_views(cams["extrinsics"], cams["intrinsics"], cams["near"], cams["far"], (H, W), bg,
                                g["means"], g["covariances"], g["harmonics"], g["opacities"])
            ((c - target) ** 2).mean().backward()
            return c, d
        with torch.no_grad():
            return render_views(cams["extrinsics"], cams["intrinsics"], cams["near"], cams["far"], (H, W), bg,
                                g["means"], g["covariances"], g["harmonics"], g["opacities"])
    for _ in range(2):
        out = step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    grads = {k: v.grad.clone() for k, v in g.items()} if train else None
    return views * steps / dt, out[0].detach().clone(), out[1].detach().clone(), grads


res = {}
for rep in range(2):
    for fast in (False, True):
        v, c, d, _ = run(fast, False, 10)
        res.setdefault("fwd_fast" if fast else "fwd_exact", []).append(round(v, 1))
        if rep == 0:
            res["img_fast" if fast else "img_exact"] = (c, d)
for fast in (False, True):
    v, c, d, gr = run(fast, True, 5)
    res["train_fast" if fast else "train_exact"] = round(v, 1)
    res["grad_fast" if fast else "grad_exact"] = gr
ce, de = res.pop("img_exact")
cf, df = res.pop("img_fast")
dpx = (ce - cf).abs().amax(dim=1)
res["color_max_abs_diff"] = float(dpx.max())
res["pixels_above_1e-4"] = int((dpx > 1e-4).sum())
res["pixels_above_1e-5"] = int((dpx > 1e-5).sum())
res["pixels"] = int(dpx.numel())
res["depth_max_abs_diff"] = float((de - df).abs().max())
ge, gf = res.pop("grad_exact"), res.pop("grad_fast")
res["grad_diff_over_max_abs"] = {k: float((ge[k] - gf[k]).abs().max() / (ge[k].abs().max() + 1e-30)) for k in ge}
print(json.dumps(res))
