#!/usr/bin/env python
"""CPU count for the depth-slab binning idea ("build it *or* show with a CPU count on C3 that it cannot
beat 5 %"): the oracle renders one config-3 view; per tile, n = its (unculled) list length and m = the position of the last
contributor of its slowest pixel (everything behind m is binned and sorted for nothing).  For slab schemes that CAN be
decided while the keys are written -- S slabs between thresholds fixed per view -- and for the unrealisable per-tile
optimum, the fraction of keys that still has to be binned + sorted, and what that is worth with the measured phase times
of the projection kernel (count + reserve + keys = 6.5 of 20.8 us per workgroup; the mask phase must run for every
instance) and of the fused sort (12.7 of 52.7 us of a workgroup's lifetime, ~8 % of its instructions).

  python profiles/tools/depth_slab_count.py [view]        (one view of c3_968x1296_1M; ~1 minute on 8 threads)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from freesplat_amd import synthetic  # noqa: E402
from util_raster import oracle_forward, view_inputs  # noqa: E402


def main(view=3):
    H, W, N = synthetic.WORKLOADS["c3_968x1296_1M"]
    scene = synthetic.make_scene(N)
    cams = synthetic.target_cameras(16)
    st = oracle_forward(view_inputs(scene, cams, view, H, W))
    gx, gy = (W + 15) // 16, (H + 15) // 16
    ranges, pl, depths, ncon = st["ranges"].astype(np.int64), st["point_list"], st["depths"], st["n_contrib"]
    n_t = ranges[:, 1] - ranges[:, 0]
    # last contributor of the slowest pixel of each tile
    pad = np.zeros((gy * 16, gx * 16), np.int64)
    pad[:H, :W] = ncon
    m_t = pad.reshape(gy, 16, gx, 16).max(axis=(1, 3)).reshape(-1)
    total = int(n_t.sum())
    print(f"view {view}: {total} instances in {len(n_t)} tiles (oracle lists, no tile cull); needed (up to the last contributor of "
          f"the slowest pixel): {int(np.minimum(m_t, n_t).sum())} = {np.minimum(m_t, n_t).sum() / total:.3f} of the keys")
    zmin_g, zmax_g = float(depths[depths > 0.2].min()), float(depths.max())
    z_sorted = [depths[pl[a:b]] for a, b in ranges]
    for S in (4, 16, 64, 256):
        edges_g = np.linspace(zmin_g, zmax_g, S + 1)[1:]                       # global slabs, uniform in z
        edges_q = np.quantile(depths[depths > 0.2], np.linspace(0, 1, S + 1)[1:])   # global slabs of equal population
        need_g = need_q = need_t = 0
        for z, n, m in zip(z_sorted, n_t, m_t):
            if n == 0:
                continue
            zl = z[min(m, n) - 1] if m > 0 else -1.0
            for edges, acc in ((edges_g, "g"), (edges_q, "q")):
                cut = edges[min(np.searchsorted(edges, zl, side="left"), S - 1)] if m > 0 else -1.0
                k = int(np.searchsorted(z, cut, side="right"))
                if acc == "g":
                    need_g += k
                else:
                    need_q += k
            # per-tile slabs (the tile's own depth range: needs zmin / zmax of the tile BEFORE its keys are written)
            if m > 0:
                e = np.linspace(z[0], z[-1], S + 1)[1:]
                need_t += int(np.searchsorted(z, e[min(np.searchsorted(e, zl, side="left"), S - 1)], side="right"))
        print(f"  S = {S:3d} slabs: keys binned + sorted  global uniform-z {need_g / total:.3f}   global equal-population "
              f"{need_q / total:.3f}   per-tile range {need_t / total:.3f}")
    f_need = np.minimum(m_t, n_t).sum() / total
    # what the unrealisable optimum is worth: projection kernel 0.082 ms, fused sort + blend 0.2015 ms per view (isolated)
    pre_bin = 0.082 * 6.5 / 20.8
    sort_lat, sort_instr = 0.2015 * 12.7 / 52.7, 0.2015 * 0.08
    for name, f in (("per-tile optimum", f_need),):
        lo = (1 - f) * (pre_bin + sort_instr)
        hi = (1 - f) * (pre_bin + sort_lat)
        print(f"  {name}: saves {1 - f:.2f} of the keys = {lo * 1e3:.0f} .. {hi * 1e3:.0f} us of the 294 us of kernels per view "
              f"({lo / 0.294:.1%} .. {hi / 0.294:.1%}; the sort phase is latency the other workgroups of a CU hide -- its "
              f"instruction share is the lower figure) BEFORE the cost of knowing the cut")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
