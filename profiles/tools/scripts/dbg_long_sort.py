import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from util_raster import small_scene, view_inputs, oracle_forward, hip_forward
from freesplat_amd import rasterizer as R
from freesplat_amd.rasterizer import debug_state
R.TILE_CULL = False
dev = torch.device("cuda:0")
H = W = 32
scene, cams = small_scene(N=6000, H=H, W=W, seed=4)
scene["covariances"] = scene["covariances"] * 400.0
scene["opacities"] = scene["opacities"] * 0.02
vi = view_inputs(scene, cams, 0, H, W)
st = oracle_forward(vi)
for mode in (False, True):
    out, leaves = hip_forward(vi, dev, requires_grad=mode)
    c = out[0].detach().cpu().numpy()
    print("requires_grad", mode, "image equal", bool((c == st["color"]).all()), "max err", float(np.abs(c - st["color"]).max()))
    if mode:
        dbg = debug_state(out[0].grad_fn.rs)
        off = dbg["offsets"].astype(np.int64)
        depths = st["depths"]
        for t in range(len(off) - 1):
            a = dbg["point_list"][off[t]:off[t + 1]].astype(np.int64); b = st["point_list"][off[t]:off[t + 1]].astype(np.int64)
            bad = np.nonzero(a != b)[0]
            print("tile", t, "n", len(a), "mismatches", len(bad), "perm", bool((np.sort(a) == np.sort(b)).all()),
                  "first", bad[:5], "a", a[bad[:5]], "b", b[bad[:5]])
            if len(bad):
                lo, hi = bad.min(), bad.max()
                exp = set(b[lo:hi + 1].tolist())
                whole = dbg["point_list"].astype(np.int64)
                found_elsewhere = sum(1 for x in whole[:off[t]].tolist() + whole[off[t + 1]:].tolist() if x in exp)
                inside = a[lo:hi + 1]; valid = inside[(inside >= 0) & (inside < len(depths))]
                print("  bad range", lo, hi, "valid ids in range", len(valid), "of", hi - lo + 1, "expected ids present in range", len(set(valid.tolist()) & exp))
                a = np.clip(a, 0, len(depths) - 1)
                da = depths[a]; print("  depth sorted?", bool((np.diff(da) >= 0).all()), "n descents", int((np.diff(da) < 0).sum()),
                                     "descents at", np.nonzero(np.diff(da) < 0)[0][:10])
                dup = len(a) - len(np.unique(a)); print("  duplicates", dup)
