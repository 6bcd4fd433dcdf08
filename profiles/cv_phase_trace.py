#!/usr/bin/env python
"""Where a wavefront of the cost-volume sweep spends its time (debug build of the library with -DFS_CV_TRACE:
  make -C freesplat_amd/csrc clean && make -C freesplat_amd/csrc EXTRA=-DFS_CV_TRACE
or a separate .so selected with FREESPLAT_LIB).  Prints mean shader cycles per (pixel group, plane) -- 32 pixels in the K = 1
sweep and in the backward, 16 in the K >= 2 sweep -- in the gather phase (plane depth -> projection -> bilinear taps ->
reduction) and in the MLP phase (K = 1: 16 MFMAs 32x32x2; K >= 2: 42 MFMAs 16x16x4; + glue)."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden"))
import numpy as np
import torch

import inputs
from freesplat_amd import _lib
from freesplat_amd.cost_volume import AVGFeatureVolumeManager


def main(V=2, K=1, h4=96, w4=128, D=128):
    dev = torch.device("cuda:0")
    L = C.CDLL(_lib.LIB_PATH)
    n = 16384 * 4
    buf = (C.c_ulonglong * n)()
    torch.manual_seed(0)
    m = AVGFeatureVolumeManager(h4, w4, num_depth_bins=D, mlp_channels=[202, 32, 32, 1], matching_dim_size=48).to(dev)
    kw = {k: v.to(dev) for k, v in inputs.cv_inputs(V, K, h4, w4, 48, seed=1).items()}
    with torch.no_grad():
        for _ in range(3):
            m(**kw)
        torch.cuda.synchronize()
        L.fs_debug_cv_trace(buf, 1)
        m(**kw)
        torch.cuda.synchronize()
        L.fs_debug_cv_trace(buf, 0)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).astype(np.float64)
    raw = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4)
    raw = raw[raw[:, 3] > 0]
    a = a[a[:, 3] > 0]
    planes = a[:, 2].sum()
    ticks, wall = (raw[:, 3] >> np.uint64(32)).astype(np.float64), (raw[:, 3] & np.uint64(0xffffffff)).astype(np.float64)
    out = {"config": f"V={V} K={K} {h4}x{w4} D={D}", "wavefronts": int(len(a)), "planes_per_wavefront": float(a[:, 2].mean()),
           "gather_cycles_per_plane": float(a[:, 0].sum() / planes), "mlp_cycles_per_plane": float(a[:, 1].sum() / planes),
           "pixels_per_group": 32 if K == 1 else 16, "mfma_cycles_per_plane": 16 * 64 if K == 1 else 42 * 32,
           "s_memtime_ticks_per_us": float(ticks.sum() / (wall.sum() / 100.0)),
           "mean_wavefront_us": float(wall.mean() / 100.0)}
    print(json.dumps(out))


def backward(V=2, K=1, h4=96, w4=128, D=128):
    """The same for the backward kernel (fs_debug_cvb_trace): forward recompute / MLP backward with the weight-gradient
    outer products (the two-phase LDS tiles interleave them; the third stamp follows at once) / feature gradients (re-gather
    for K > 1 + scatter), shader cycles per (32-pixel group, plane)."""
    dev = torch.device("cuda:0")
    L = C.CDLL(_lib.LIB_PATH)
    n = 16384 * 6
    buf = (C.c_ulonglong * n)()
    torch.manual_seed(0)
    m = AVGFeatureVolumeManager(h4, w4, num_depth_bins=D, mlp_channels=[202, 32, 32, 1], matching_dim_size=48).to(dev)
    kw = {k: v.to(dev) for k, v in inputs.cv_inputs(V, K, h4, w4, 48, seed=1).items()}
    kw = {k: (v.clone().requires_grad_(True) if k in ("cur_feats", "src_feats") else v) for k, v in kw.items()}
    for i in range(3):
        if i == 2:
            torch.cuda.synchronize()
            L.fs_debug_cvb_trace(buf, 1)
        o = m(**kw)
        o.backward(torch.ones_like(o))
    torch.cuda.synchronize()
    L.fs_debug_cvb_trace(buf, 0)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 6).astype(np.float64)
    a = a[a[:, 4] > 0]
    planes = a[:, 4].sum()
    names = ["forward_recompute", "mlp_backward", "weight_gradients", "feature_gradients"]
    out = {"config": f"backward V={V} K={K} {h4}x{w4} D={D}", "wavefronts": int(len(a)), "planes_per_wavefront": float(a[:, 4].mean())}
    out.update({nm + "_cycles_per_plane": float(a[:, i].sum() / planes) for i, nm in enumerate(names)})
    out["total_cycles_per_plane"] = float(a[:, 5].sum() / planes)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
    main(V=3, K=2, h4=242, w4=324)
    main(V=10, K=8)
    backward()
    backward(V=3, K=2, h4=242, w4=324, D=64)
