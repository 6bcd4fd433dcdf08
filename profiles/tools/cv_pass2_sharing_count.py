#!/usr/bin/env python
"""CPU count behind "the K sources of one tile read different pixels" (cost volume, "Pass 2, round 5"): for the
benchmarked cameras, per (current view, plane, 8 x 8 source tile): the preimage box of the tile in the current view for each of
the K sources (the inverse-plane-homography box cv_src_grad_kernel walks), and
   sum over the sources of the box pixels  /  pixels of the UNION of the K boxes
= how many times a record would be reused out of a cache shared by the K workgroups of one tile IF they walked the planes in
lockstep (K = every source reads the same pixels, 1 = disjoint boxes).  Also the records one view has in flight when all its
tiles sweep one group of four planes, against the 4 MB L2 of an XCD.   python profiles/tools/cv_pass2_sharing_count.py [fvt10|c3scale]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import inputs  # noqa: E402

CFG = {"c3scale": (3, 2, 242, 324, 128), "fvt10": (10, 8, 96, 128, 128)}


def main(name, TW=8, TH=8, step=8):
    V, K, h, w, D = CFG[name]
    kw = inputs.cv_inputs(V, K, h, w, 48, seed=1)
    P = (kw["src_Ks"] @ kw["src_extrinsics"])[:, :, :3].numpy().astype(np.float64)
    iK = kw["cur_invK"][:, :3, :3].numpy().astype(np.float64)
    planes = 1.0 / (1 / 0.5 + (1 / 15.0 - 1 / 0.5) * np.linspace(0, 1, D))
    tiles_x, tiles_y = (w + TW - 1) // TW, (h + TH - 1) // TH
    tot_sum = tot_union = 0
    per_plane = {}
    for b in range(V):
        for d in range(0, D, step):
            Gi = [np.linalg.inv(planes[d] * P[b, k][:, :3] @ iK[b] + np.outer(P[b, k][:, 3], [0, 0, 1.0])) for k in range(K)]
            s_sum = s_union = 0
            for ty in range(tiles_y):
                for tx in range(tiles_x):
                    x0, y0 = tx * TW, ty * TH
                    x1, y1 = min(w, x0 + TW), min(h, y0 + TH)
                    cs = np.array([[x0 - 0.55, y0 - 0.55, 1], [x1 + 0.55, y0 - 0.55, 1], [x0 - 0.55, y1 + 0.55, 1], [x1 + 0.55, y1 + 0.55, 1]]).T
                    cover = np.zeros((h, w), bool)
                    for k in range(K):
                        pre = Gi[k] @ cs
                        c = pre[2]
                        if not (c > 0).all():
                            continue
                        uu, vv = pre[0] / c - 0.5, pre[1] / c - 0.5
                        bx0, bx1 = max(0, int(np.ceil(uu.min() - 0.05))), min(w - 1, int(np.floor(uu.max() + 0.05)))
                        by0, by1 = max(0, int(np.ceil(vv.min() - 0.05))), min(h - 1, int(np.floor(vv.max() + 0.05)))
                        if bx1 < bx0 or by1 < by0:
                            continue
                        s_sum += (bx1 - bx0 + 1) * (by1 - by0 + 1)
                        cover[by0:by1 + 1, bx0:bx1 + 1] = True
                    s_union += int(cover.sum())
            tot_sum += s_sum
            tot_union += s_union
            per_plane.setdefault(d, [0, 0])
            per_plane[d][0] += s_sum
            per_plane[d][1] += s_union
    rec = (48 + 2) * 4
    print(f"{name}: {V} views, K = {K}, {h}x{w}, tiles {TW}x{TH}; planes sampled every {step}")
    print(f"  box pixels summed over the K sources / pixels of the union of the K boxes of a tile: {tot_sum / max(tot_union, 1):.2f} "
          f"(K = {K} would be full sharing)")
    print("  by plane (near -> far): " + ", ".join(f"d={d}: {a / max(u, 1):.2f}" for d, (a, u) in sorted(per_plane.items())))
    print(f"  records of ONE view in flight when its tiles sweep one group of 4 planes: {4 * h * w * rec / 1e6:.1f} MB "
          f"(L2 of an XCD: 4 MB; all planes in flight, i.e. no lockstep: {D * h * w * rec / 1e6:.0f} MB per view)")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "fvt10")
