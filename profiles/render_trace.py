#!/usr/bin/env python
"""Per-tile timeline of render_kernel (debug build only):
    make -C freesplat_amd/csrc clean && make -C freesplat_amd/csrc EXTRA=-DFS_RENDER_TRACE
    python profiles/render_trace.py gpurun_out/render_trace.json
Each workgroup stamps wall_clock64() (100 MHz) at entry and when its last / first wavefront leaves; the script
renders one C3 view and reports the concurrency profile and what a perfectly balanced schedule would take."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freesplat_amd import _lib, synthetic  # noqa: E402
from freesplat_amd.decoder import render_views  # noqa: E402


def main(out):
    dev = torch.device("cuda:0")
    H, W, N = synthetic.WORKLOADS["c3_968x1296_1M"]
    scene = synthetic.make_scene(N)
    cams = {k: v.to(dev) for k, v in synthetic.target_cameras(8).items()}
    g = {k: scene[k].to(dev) for k in ("means", "covariances", "harmonics", "opacities")}
    one = {k: v[:1] for k, v in cams.items()}
    L = _lib.lib()
    L.fs_debug_render_trace.restype = C.c_int
    with torch.no_grad():
        for _ in range(3):
            render_views(one["extrinsics"], one["intrinsics"], one["near"], one["far"], (H, W), torch.zeros(1, 3, device=dev),
                         g["means"], g["covariances"], g["harmonics"], g["opacities"])
        torch.cuda.synchronize()
        L.fs_debug_render_trace(None, 1)
        render_views(one["extrinsics"], one["intrinsics"], one["near"], one["far"], (H, W), torch.zeros(1, 3, device=dev),
                     g["means"], g["covariances"], g["harmonics"], g["opacities"])
        torch.cuda.synchronize()
    buf = np.zeros(4 * 8192, np.uint64)
    L.fs_debug_render_trace(buf.ctypes.data_as(C.c_void_p), 0)
    t = buf.reshape(8192, 4)
    live = t[:, 0] > 0
    t = t[live]
    t0 = t[:, 0].min()
    beg = (t[:, 0] - t0).astype(np.float64) / 100.0          # us
    end = (t[:, 1] - t0).astype(np.float64) / 100.0
    first_end = ((~t[:, 3]) - t0).astype(np.float64) / 100.0
    n = (t[:, 2] >> np.uint64(32)).astype(np.int64)
    hw = (t[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    dur = end - beg
    span = end.max()
    grid = np.linspace(0, span, 41)
    conc = [(int(((beg <= x) & (end > x)).sum())) for x in grid]
    res = dict(tiles=int(live.sum()), span_us=float(span), sum_block_us=float(dur.sum()),
               mean_block_us=float(dur.mean()), p50=float(np.median(dur)), p90=float(np.percentile(dur, 90)),
               max_block_us=float(dur.max()), mean_first_wave_leaves_frac=float(((first_end - beg) / np.maximum(dur, 1e-9)).mean()),
               concurrency_over_time=conc, corr_dur_vs_list_len=float(np.corrcoef(dur, n)[0, 1]),
               mean_list_len=float(n.mean()), max_list_len=int(n.max()),
               last_start_us=float(beg.max()), blocks_started_after_half=int((beg > span / 2).sum()))
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))
    np.savez_compressed(out.replace(".json", ".npz"), beg=beg, end=end, first_end=first_end, n=n, hw=hw)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/render_trace.json")
