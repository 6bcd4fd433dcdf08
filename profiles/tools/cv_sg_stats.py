"""Counters of the source-tile sweep (debug build: make -C freesplat_amd/csrc VARIANT=sgstats EXTRA=-DFS_CV_SG_STATS):
   FREESPLAT_LIB=freesplat_amd/libfreesplat_hip_sgstats.so python profiles/tools/cv_sg_stats.py [workload ...]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402

import inputs  # noqa: E402
from freesplat_amd import _lib  # noqa: E402
from freesplat_amd.cost_volume import AVGFeatureVolumeManager  # noqa: E402

WL = {"native_K1": (2, 1, 96, 128), "c3scale_K2": (3, 2, 242, 324), "fvt10_K8": (10, 8, 96, 128), "small": (3, 2, 48, 64),
      "oblique": (3, 2, 30, 40)}     # (one view turned by 1.2 rad: tiles that straddle the planes' horizon walk the whole image)
L = _lib.lib()
dev = torch.device("cuda:0")
for name in (sys.argv[1:] or ["small", "native_K1", "fvt10_K8"]):
    V, K, h4, w4 = WL[name]
    D, C = (16 if name == "oblique" else 128), 48
    torch.manual_seed(0)
    m = AVGFeatureVolumeManager(h4, w4, num_depth_bins=D, mlp_channels=[202, 32, 32, 1], matching_dim_size=C).to(dev)
    a = {k: v.to(dev) for k, v in inputs.cv_inputs(V, K, h4, w4, C, seed=34 if name == "oblique" else 1,
                                                    oblique=1.2 if name == "oblique" else 0.0).items()}
    a["cur_feats"].requires_grad_(True); a["src_feats"].requires_grad_(True)
    o = m(**a)
    buf = (ctypes.c_ulonglong * 8)()
    L.fs_debug_cv_sg_stats(buf, 1)
    o.backward(torch.ones_like(o))
    torch.cuda.synchronize()
    L.fs_debug_cv_sg_stats(buf, 0)
    s = list(buf)
    pps = V * K * D * h4 * w4
    print(f"{name}: (pixel, plane, source) {pps}; wave iterations {s[0]} (x64 = {64 * s[0] / pps:.2f} per pps), pixels with a tap in the tile "
          f"{s[1]} ({s[1] / pps:.2f}); cells walked {s[2]}, fallbacks {s[3]}, skipped behind {s[4]}, box pixels {s[5]} ({s[5] / pps:.2f}); "
          f"claim rounds {s[6]} ({s[6] / max(s[0], 1):.2f} per iteration)", flush=True)
