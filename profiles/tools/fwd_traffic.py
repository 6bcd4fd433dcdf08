#!/usr/bin/env python
"""HBM traffic per unit of work (PMC FETCH_SIZE / WRITE_SIZE) of every forward workload bench.py reports a roofline for.

  python profiles/tools/fwd_traffic.py run <workload> [calls]      exactly 1 warm + `calls` inference calls, nothing else
  bash   profiles/tools/fwd_traffic.sh <tag>                       rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the above
                                                                   for all workloads (each counter in its own run)
  python profiles/tools/fwd_traffic.py summarize <tag>             -> profiles/<tag>_traffic.json

Per workload: bytes of ALL library kernels (fs::*) of the run / (calls + 1), split per kernel (bytes per call and mean
per launch).  read = 2 * FETCH_SIZE * 1024 (gfx950 correction, MI355X_MICROARCH.md "HBM"), write = WRITE_SIZE * 1024.
Units: raster_* one rendered VIEW (a call renders 16); cv_* one cost-volume call (all current views); ptf_* one fold."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]

WORKLOADS = {   # name -> (units per call, default calls)
    "raster_c3": (16, 4), "raster_c2": (16, 4), "raster_closeup": (4, 3), "cv_native_K1": (1, 6), "cv_c3scale_K2": (1, 3), "cv_fvt10_K8": (1, 3),
    "cv_fvt10_K8_cl": (1, 3),     # the same call on channels_last feature maps (read in place: no re-layout pass)
    "ptf_2_views": (1, 6), "ptf_10_views": (1, 3), "ptf_3_views": (1, 3),      # (ptf_3_views: 3 views at 968x1296)
    # training steps of the cost volume (forward + backward w.r.t. features and MLP): unit = one step
    "cvt_native_K1": (1, 4), "cvt_c3scale_K2": (1, 2), "cvt_fvt10_K8": (1, 2),
}


def run(name, calls):
    import torch
    dev = torch.device("cuda:0")
    if name.startswith("raster"):
        from freesplat_amd import synthetic
        from freesplat_amd.decoder import render_views
        wl = {"raster_c3": "c3_968x1296_1M", "raster_c2": "c2_640x480_300k", "raster_closeup": "c3_closeup_968x1296_1M"}[name]
        H, W, N = synthetic.WORKLOADS[wl]
        scene = synthetic.workload_scene(wl)
        nv = WORKLOADS[name][0]
        cams = {k: v.to(dev) for k, v in synthetic.target_cameras(nv).items()}
        g = {k: scene[k].to(dev) for k in ("means", "covariances", "harmonics", "opacities")}
        bg = torch.zeros(nv, 3, device=dev)
        if name == "raster_closeup":     # (no overflow + re-render inside the traced calls: start from the capacity the bench ends up with)
            from freesplat_amd import rasterizer as R
            R._state(dev).note_overflow(24_000_000, 16_000, H, W)
        fn = lambda: render_views(cams["extrinsics"], cams["intrinsics"], cams["near"], cams["far"], (H, W), bg, g["means"],
                                  g["covariances"], g["harmonics"], g["opacities"])
    elif name.startswith("cv"):
        import inputs
        from freesplat_amd.cost_volume import AVGFeatureVolumeManager
        V, K, h4, w4 = {"native_K1": (2, 1, 96, 128), "c3scale_K2": (3, 2, 242, 324), "fvt10_K8": (10, 8, 96, 128)}[name.split("_", 1)[1].replace("_cl", "")]
        torch.manual_seed(0)
        m = AVGFeatureVolumeManager(matching_height=h4, matching_width=w4, num_depth_bins=128, mlp_channels=[202, 32, 32, 1],
                                    matching_dim_size=48).to(dev)
        args = {k: v.to(dev) for k, v in inputs.cv_inputs(V, K, h4, w4, 48, seed=1).items()}
        if name.endswith("_cl"):
            args["cur_feats"] = args["cur_feats"].contiguous(memory_format=torch.channels_last)
            args["src_feats"] = args["src_feats"].permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3)
        fn = lambda: m(**args)
        if name.startswith("cvt_"):
            args["cur_feats"].requires_grad_(True)
            args["src_feats"].requires_grad_(True)
            for _ in range(calls + 1):
                o = m(**args)
                o.backward(torch.ones_like(o))
            torch.cuda.synchronize()
            return
    else:
        from test_ptf_hip import _scene
        from freesplat_amd.ptf import PixelwiseTripletFusion
        V = {"ptf_2_views": 2, "ptf_10_views": 10, "ptf_3_views": 3}[name]
        h, w = (968, 1296) if name == "ptf_3_views" else (384, 512)
        E, Kn, depths, lat, dens, wts, coords = _scene(V, h, w, seed=5)
        torch.manual_seed(1)
        m = PixelwiseTripletFusion().to(dev)
        d = lambda t: t.to(dev)
        a = ([d(lat)], [d(coords)], d(dens), d(wts), d(depths), d(E)[None], d(Kn)[None], (h, w))
        fn = lambda: m.fuse_gaussians(*a)
    with torch.no_grad():
        for _ in range(calls + 1):
            fn()
    torch.cuda.synchronize()


def summarize(tag):
    out = {"note": __doc__.split("\n\n")[1].replace("\n", " "), "workloads": {}}
    for name, (units, _) in WORKLOADS.items():
        src = os.path.join(ROOT, "gpurun_out", f"traffic_{tag}", name)
        meta = os.path.join(src, "calls.txt")
        if not os.path.exists(meta):
            continue
        calls = int(open(meta).read().strip()) + 1
        per = collections.defaultdict(lambda: {"read": 0.0, "write": 0.0, "launches": 0})
        for sub, key, scale in (("fetch", "read", 2048.0), ("write", "write", 1024.0)):
            hits = glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True)
            if not hits:
                continue
            for r in csv.DictReader(open(hits[0])):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if "fs::" not in k:
                    continue
                per[k][key] += float(r["Counter_Value"]) * scale
                if sub == "fetch":
                    per[k]["launches"] += 1
        n = calls * units
        tot_r, tot_w = sum(v["read"] for v in per.values()), sum(v["write"] for v in per.values())
        out["workloads"][name] = {
            "unit": "view" if name.startswith("raster") else ("step" if name.startswith("cvt") else ("call" if name.startswith("cv") else "fold")),
            "units_measured": n, "hbm_bytes_per_unit": (tot_r + tot_w) / n, "read_bytes_per_unit": tot_r / n,
            "write_bytes_per_unit": tot_w / n,
            "kernels": {k: {"bytes_per_unit": (v["read"] + v["write"]) / n, "launches_per_unit": v["launches"] / n,
                            "bytes_per_launch": (v["read"] + v["write"]) / max(v["launches"], 1)} for k, v in sorted(per.items())}}
    path = os.path.join(ROOT, "profiles", f"{tag}_traffic.json")
    if os.path.exists(path):
        # a run over SOME workloads must not drop the others from the tracked file (bench.py reads its traffic rows from it:
        # a partial rewrite on the GPU box once left the cost-volume backward rows of a bench line empty)
        old = json.load(open(path)).get("workloads", {})
        out["workloads"] = {**old, **out["workloads"]}
    json.dump(out, open(path, "w"), indent=1)
    for k, v in out["workloads"].items():
        print(f"{k:16s} {v['hbm_bytes_per_unit'] / 1e6:9.1f} MB per {v['unit']}  (rd {v['read_bytes_per_unit'] / 1e6:.1f}, wr {v['write_bytes_per_unit'] / 1e6:.1f})")


_LIB_BYTES = None


def kernel_shipped(name):
    """Whether the library still contains a kernel of this (demangled) name: its identifier appears in the mangled symbol
    inside the embedded code object.  A traffic figure of a kernel that no longer exists must not be reported."""
    global _LIB_BYTES
    if _LIB_BYTES is None:
        from freesplat_amd import _lib
        _LIB_BYTES = open(_lib.LIB_PATH, "rb").read()
    ident = name.split("<")[0].split("::")[-1].strip()
    return f"{len(ident)}{ident}".encode() in _LIB_BYTES


def lookup(workload, kernel_prefix=None):
    """(bytes, source file) of `workload` from the newest committed profiles/*_traffic.json whose kernels ALL still exist
    in the library (a file that names a deleted kernel is skipped for that workload): per unit, or per launch of the kernel
    whose name starts with `kernel_prefix`.  (None, None) when absent."""
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
        try:
            w = json.load(open(f))["workloads"].get(workload)
        except Exception:
            continue
        if not w:
            continue
        if not all(kernel_shipped(k) for k in w["kernels"]):
            continue
        if kernel_prefix is None:
            return float(w["hbm_bytes_per_unit"]), os.path.relpath(f, ROOT)
        for k, v in w["kernels"].items():
            if k.startswith(kernel_prefix):
                return float(v["bytes_per_launch"]), os.path.relpath(f, ROOT)
    return None, None


if __name__ == "__main__":
    if sys.argv[1] == "run":
        name = sys.argv[2]
        run(name, int(sys.argv[3]) if len(sys.argv) > 3 else WORKLOADS[name][1])
    else:
        summarize(sys.argv[2])
