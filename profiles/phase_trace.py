#!/usr/bin/env python
"""Where the time of a workgroup goes inside preprocess (projection + binning) / sort_blend (debug build only):
    make -C freesplat_amd/csrc VARIANT=trace EXTRA=-DFS_PHASE_TRACE
    FREESPLAT_LIB=$PWD/freesplat_amd/libfreesplat_hip_trace.so python profiles/phase_trace.py
FS_PT(kernel, k) stamps wall_clock64() (100 MHz) at phase boundaries for thread 0 of the first 4096 workgroups;
this renders one C3 view and prints the mean / median duration of every phase."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freesplat_amd import _lib, synthetic  # noqa: E402
from freesplat_amd.decoder import render_views  # noqa: E402

PHASES = {
    0: ("preprocess (projection + binning)", ["stage inputs + sync", "project / conic / SH", "write records + sync", "quadrant masks",
                       "workgroup box", "zero + count (LDS atomics)", "reserve slots (returning global atomics)",
                       "rank + write keys"]),
    2: ("sort_blend (wavefront 0)", ["load keys + LDS bucket sort", "wait for the other wavefronts", "blend quadrant 0"]),
}


def main():
    dev = torch.device("cuda:0")
    H, W, N = synthetic.WORKLOADS["c3_968x1296_1M"]
    scene = synthetic.make_scene(N)
    cams = {k: v[:1].to(dev) for k, v in synthetic.target_cameras(8).items()}
    g = {k: scene[k].to(dev) for k in ("means", "covariances", "harmonics", "opacities")}
    L = _lib.lib()
    L.fs_debug_phase_trace.restype = C.c_int
    buf = np.zeros(4 * 4096 * 10, np.uint64)
    with torch.no_grad():
        for it in range(4):
            render_views(cams["extrinsics"], cams["intrinsics"], cams["near"], cams["far"], (H, W), torch.zeros(1, 3, device=dev),
                         g["means"], g["covariances"], g["harmonics"], g["opacities"])
            torch.cuda.synchronize()
            L.fs_debug_phase_trace(buf.ctypes.data_as(C.c_void_p))     # (copies out and clears)
    t = buf.reshape(4, 4096, 10).astype(np.float64) / 100.0
    for kern, (name, phases) in PHASES.items():
        n = len(phases) + 1
        x = t[kern][:, :n]
        x = x[(x > 0).all(axis=1)]
        if not len(x):
            print(f"{name}: no samples")
            continue
        d = np.diff(x, axis=1)
        print(f"{name}: {len(x)} workgroups, {(x[:, -1] - x[:, 0]).mean():.2f} us each")
        for ph, m, md in zip(phases, d.mean(0), np.median(d, 0)):
            print(f"   {ph:34s} mean {m:6.2f} us   median {md:6.2f}")


if __name__ == "__main__":
    main()
