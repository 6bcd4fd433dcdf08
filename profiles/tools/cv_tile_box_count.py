#!/usr/bin/env python
"""CPU count for the source-tile form of the cost-volume backward (round 4): for every (current view, source, plane,
source tile of TW x TH texels) the preimage box of the tile in the current view -- the bounding box of the tile
rectangle's corners (grown by the bilinear footprint) mapped through the INVERSE plane homography -- is compared with the
exact set of current pixels that have a bilinear tap inside the tile (forward projection, float32, the kernel's op order).
Reports: coverage (must be exact: no pixel with a tap in the tile outside the box), pixel visits per useful (pixel, plane,
source) = the halo overhead of the scatter pass, fallbacks (tiles whose corners straddle the plane's horizon), and the
atomic-record count of the round-3 scatter for comparison.

  python profiles/tools/cv_tile_box_count.py [native|c3scale|fvt10|behind]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import inputs  # noqa: E402

CFG = {"native": (2, 1, 96, 128, 128, False), "c3scale": (3, 2, 242, 324, 128, False), "fvt10": (10, 8, 96, 128, 128, False),
       "behind": (3, 2, 15, 21, 6, True), "small": (3, 2, 48, 64, 32, False)}


def forward_taps(G, h, w):
    """ix, iy, qz per pixel for one (view, source, plane): float32, kernel op order (approximately)."""
    f = np.float32
    v, u = np.meshgrid(np.arange(h, dtype=f), np.arange(w, dtype=f), indexing="ij")
    up, vp = u + f(0.5), v + f(0.5)
    q = G[:, 0, None, None] * up + G[:, 1, None, None] * vp + G[:, 2, None, None]
    qz = q[2]
    zz = qz + f(1e-8)
    sc = np.where(np.abs(qz) > 1e-8, f(1) / zz, f(1))
    ix = ((f(2) * (q[0] * sc) * f(1.0 / w) - f(1) + f(1)) * f(w) - f(1)) * f(0.5)
    iy = ((f(2) * (q[1] * sc) * f(1.0 / h) - f(1) + f(1)) * f(h) - f(1)) * f(0.5)
    return ix, iy, zz


def main(name, TW=16, TH=8, planes_step=None):
    V, K, h, w, D, behind = CFG[name]
    planes_step = planes_step or (1 if D <= 32 else 8)
    kw = inputs.cv_inputs(V, K, h, w, 48, seed=1, behind=behind)
    P = (kw["src_Ks"] @ kw["src_extrinsics"])[:, :, :3].numpy().astype(np.float64)
    iK = kw["cur_invK"][:, :3, :3].numpy().astype(np.float64)
    ramp = np.linspace(0, 1, D)
    planes = 1.0 / (1 / 0.5 + (1 / 15.0 - 1 / 0.5) * ramp)
    tiles_x, tiles_y = (w + TW - 1) // TW, (h + TH - 1) // TH
    tot_visits = tot_useful = tot_fallback = tot_skipped = tot_cells = missed = 0
    tot_box_px = 0
    for b in range(V):
        for k in range(K):
            for d in range(0, D, planes_step):
                G = planes[d] * P[b, k][:, :3] @ iK[b] + np.outer(P[b, k][:, 3], [0, 0, 1.0])
                Gi = np.linalg.inv(G)
                ix, iy, zz = forward_taps(G.astype(np.float32), h, w)
                fx0, fy0 = np.floor(ix), np.floor(iy)
                front = zz > 0
                for ty in range(tiles_y):
                    for tx in range(tiles_x):
                        x0, y0 = tx * TW, ty * TH
                        x1, y1 = min(w, x0 + TW), min(h, y0 + TH)
                        # pixels with a tap inside [x0,x1) x [y0,y1): floor in [x0-1, x1-1]
                        hit = front & (fx0 >= x0 - 1) & (fx0 <= x1 - 1) & (fy0 >= y0 - 1) & (fy0 <= y1 - 1)
                        tot_cells += 1
                        # box from the inverse homography of the grown rectangle (X = ix + 0.5)
                        m = 0.05
                        cs = np.array([[x0 - 0.5 - m, y0 - 0.5 - m, 1], [x1 + 0.5 + m, y0 - 0.5 - m, 1],
                                       [x0 - 0.5 - m, y1 + 0.5 + m, 1], [x1 + 0.5 + m, y1 + 0.5 + m, 1]]).T
                        pre = Gi @ cs
                        c = pre[2]
                        if (c < 0).all():
                            tot_skipped += 1
                            if hit.any():
                                missed += int(hit.sum())
                            continue
                        if (c > 0).all() and c.min() > 1e-3 * c.max():
                            uu, vv = pre[0] / c - 0.5, pre[1] / c - 0.5
                            bx0, bx1 = max(0, int(np.ceil(uu.min() - 0.05))), min(w - 1, int(np.floor(uu.max() + 0.05)))
                            by0, by1 = max(0, int(np.ceil(vv.min() - 0.05))), min(h - 1, int(np.floor(vv.max() + 0.05)))
                        else:
                            bx0, bx1, by0, by1 = 0, w - 1, 0, h - 1
                            tot_fallback += 1
                        if bx1 < bx0 or by1 < by0:
                            if hit.any():
                                missed += int(hit.sum())
                            continue
                        inside = np.zeros_like(hit)
                        inside[by0:by1 + 1, bx0:bx1 + 1] = True
                        missed += int((hit & ~inside).sum())
                        npx = (bx1 - bx0 + 1) * (by1 - by0 + 1)
                        tot_box_px += npx
                        tot_visits += (npx + 63) // 64 * 64
                        tot_useful += int(hit.sum())
                pass
    nd = len(range(0, D, planes_step))
    pix_plane_src = V * K * nd * h * w
    print(f"{name}: tile {TW}x{TH}; (pixel, plane, source) = {pix_plane_src}; pixels with a tap in some tile (sum over tiles) = "
          f"{tot_useful} ({tot_useful / pix_plane_src:.2f} per (pixel, plane, source)); box pixels {tot_box_px} "
          f"({tot_box_px / pix_plane_src:.2f}x), visits rounded to 64-pixel iterations {tot_visits} ({tot_visits / pix_plane_src:.2f}x); "
          f"cells {tot_cells}, skipped (behind) {tot_skipped}, fallbacks {tot_fallback}; MISSED {missed}")


if __name__ == "__main__":
    nm = sys.argv[1] if len(sys.argv) > 1 else "native"
    tw = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    th = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    main(nm, tw, th)
