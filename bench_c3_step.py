#!/usr/bin/env python
"""BASELINE config 3 AS IT IS WRITTEN -- "ScanNet 3-views, 968x1296, ~1.0M Gaussians, fwd+bwd training step, 1xMI355X" -- as ONE
composed step, not as isolated pieces:

    encoder_forward (what compat.patch_reference() binds as EncoderFreeSplat.forward, encoder_freesplat.py:196-429)
      backbone*  ->  HIP cost volume (242x324, K = 2, D = 128)  ->  cv_encoder*  ->  depth decoder trunk*  ->  HIP depth tail
      ->  HIP unprojection  ->  HIP PTF fold (3 x 968x1296 = 3.76 M raw Gaussians)  ->  to_gaussians*  ->  HIP Gaussian head
    DecoderSplattingCUDA on 4 target views (decoder_splatting_cuda.py:35-75)  ->  MSE  ->  backward through everything
    (the loop of src/model/model_wrapper.py:227-303)

* = out-of-scope modules of the reference (EfficientNet backbone, cost-volume encoder, depth-decoder
convolutions, skip convolution, the latent -> raw-Gaussian linear layer): small torch stand-ins of the right SHAPES, as in
tests/test_composed_dropin.py, so that every hot-path stage sees config 3's tensor sizes (bias-free here: a bias add / a bias
gradient is a torch elementwise / reduction kernel that the kernel trace could not tell from the hot path's own glue).  Their own time is measured apart
(the same modules run alone, forward + backward, on the captured inputs) and is not the subject.

Reported (one stream, HIP events): ms per whole step; ms of library kernels per stage (fs_profile_* hooks: cost_volume,
encoder_tail, ptf, preprocess, tile_scan, render, render_bwd, preprocess_bwd); the rest (stand-in modules + glue + idle).  The
GLUE between the stages (reshapes that copy, index kernels, fills, reductions, the loss) is separated from the stand-ins' own
kernels in a rocprofv3 kernel trace of the same step, by kernel name: profiles/tools/c3_step_glue.py.
`python bench_c3_step.py [--steps 3]` prints the section as one JSON line; bench.py embeds it as `c3_train_step_hotpath`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import torch
from torch import nn

NEAR, FAR = 0.5, 15.0


def _up2(x):
    return torch.nn.functional.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)


class _Backbone(nn.Module):
    """[V,3,h,w] -> [ [V,8,h/2,w/2], [V,C,h/4,w/4] ]  (the two levels encoder_forward reads)"""

    def __init__(self, C):
        super().__init__()
        self.c0 = nn.Conv2d(3, 8, 3, stride=2, padding=1, bias=False)
        self.c1 = nn.Conv2d(8, C, 3, stride=2, padding=1, bias=False)

    def forward(self, x):
        f0 = torch.tanh(self.c0(x))
        return [f0, self.c1(f0)]


class _CVEncoder(nn.Module):
    def __init__(self, D, C):
        super().__init__()
        self.c = nn.Conv2d(D + C, 24, 3, padding=1, bias=False)

    def forward(self, volume, feats):
        return [torch.tanh(self.c(torch.cat([volume, feats[0]], 1)))]


class _DepthTrunk(nn.Module):
    """The depth decoder's convolutions (stand-in): plane logits at half resolution, 1 + 64 head channels at full."""

    def __init__(self, D):
        super().__init__()
        self.conv_depth = nn.Conv2d(8 + 24, D, 3, padding=1, bias=False)
        self.conv_last = nn.Conv2d(8 + 24, 1 + 64, 3, padding=1, bias=False)

    def forward(self, f0, f1):
        x = torch.cat([f0, _up2(f1)], 1)
        return self.conv_depth(x), self.conv_last(_up2(x))


class _DepthDecoder(nn.Module):
    def __init__(self, D):
        super().__init__()
        import numpy as np
        from freesplat_amd.depth_tail import depth_regression_tail
        self.max_depth = 1
        self.trunk = _DepthTrunk(D)
        self.tail = lambda lg, cd: depth_regression_tail(lg, cd, True, True)
        self.register_buffer("cand", torch.linspace(np.log(1.2), np.log(2.6), D))

    def forward(self, feats):
        logits, head = self.trunk(feats[0], feats[1])
        r = self.tail(logits, self.cand)
        return {"depth_pred_s0_b1hw": r["depth"], "log_depth_pred_s0_b1hw": r["coarse"], "depth_pred_s-1_b1hw": r["depth_map"],
                "depth_weights": r["depth_weights"], "output_pred_s-1_b1hw": head}


class _Encoder(nn.Module):
    """The attributes encoder_forward reads from an EncoderFreeSplat (encoder_freesplat.py:100-188), hot-path members = HIP."""

    def __init__(self, H, W, V, D, C, num_views=None):
        super().__init__()
        from freesplat_amd.cost_volume import AVGFeatureVolumeManager
        from freesplat_amd.gaussian_adapter import GaussianAdapter, GaussianAdapterCfg
        from freesplat_amd.ptf import GRU
        # cfg.num_views: how many views enter one cost volume (the view itself + its pose-nearest sources, encoder_freesplat.py:236-254)
        self.cfg = types.SimpleNamespace(num_views=num_views or V, num_surfaces=1)
        self.max_depth = 1
        torch.manual_seed(11)
        self.backbone = _Backbone(C)
        self.cv_encoder = _CVEncoder(D, C)
        self.high_resolution_skip = nn.ModuleList([nn.Conv2d(3, 64, 3, padding=1, bias=False)])
        self.to_gaussians = nn.Sequential(nn.ReLU(), nn.Linear(64, 36, bias=False))
        self.cost_volume = AVGFeatureVolumeManager(H // 4, W // 4, num_depth_bins=D, mlp_channels=[202, 32, 32, 1],
                                                   matching_dim_size=C)
        self.gru = GRU()
        self.depth_decoder = _DepthDecoder(D)
        self.gaussian_adapter = GaussianAdapter(GaussianAdapterCfg(0.5, 15.0, 2))

    def fuse_gaussians(self, *a, **k):
        from freesplat_amd.ptf import fuse_gaussians
        return fuse_gaussians(self, *a, **k)


def committed_glue():
    """{glue_ms_per_step, glue_frac_of_hotpath_gpu_time, source} from the newest profiles/*_c3_step_glue.json, or None."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_c3_step_glue.json")), reverse=True):
        try:
            d = json.load(open(f))
            return {"glue_ms_per_step": d["per_step_ms"]["glue"], "glue_frac_of_hotpath_gpu_time": d["glue_frac_of_hotpath_gpu_time"],
                    "source": os.path.relpath(f, ROOT)}
        except Exception:
            continue
    return None


def bench_c3_step(dev, steps=5, warmup=3, H=968, W=1296, V=3, D=128, C=48, n_targets=4, trace_steps=0, train=True,
                  num_views=None, workload="c3_train_step_hotpath", sh_fp16=False) -> dict:
    import inputs
    from freesplat_amd import _lib
    from freesplat_amd.decoder import DecoderSplattingCUDA
    from freesplat_amd.encoder_forward import encoder_forward
    enc = _Encoder(H, W, V, D, C, num_views=num_views).to(dev)
    dec = DecoderSplattingCUDA((0.0, 0.0, 0.0)).to(dev)
    g = torch.Generator().manual_seed(99)
    E, Kn = inputs.cameras(V, H, W, baseline=0.3, seed=5)
    ctx = {"image": torch.rand(1, V, 3, H, W, generator=g).to(dev), "extrinsics": E[None].to(dev), "intrinsics": Kn[None].to(dev),
           "near": torch.full((1, V), NEAR, device=dev), "far": torch.full((1, V), FAR, device=dev)}
    tgt_E = inputs.cameras(n_targets, H, W, baseline=0.2, seed=9)[0][None].to(dev)
    tgt_K = ctx["intrinsics"][:, :1].expand(1, n_targets, 3, 3).contiguous()
    target = torch.rand(1, n_targets, 3, H, W, generator=g).to(dev)
    near_t, far_t = torch.full((1, n_targets), NEAR, device=dev), torch.full((1, n_targets), FAR, device=dev)
    info = {}

    def step():
        if not train:       # evaluation: test_step's encoder + decoder calls (model_wrapper.py:314-324), no autograd
            with torch.no_grad():
                res = encoder_forward(enc, dict(ctx), 0, is_testing=True)
                gs = res["gaussians"][0]
                if sh_fp16:      # BASELINE config 5's storage option: the SH coefficients kept in fp16 (FS_RASTER_SH_FP16)
                    import dataclasses
                    gs = dataclasses.replace(gs, harmonics=gs.harmonics.half())
                out = dec(gs, tgt_E, tgt_K, near_t, far_t, (H, W), depth_mode=None)
            info["gaussians"] = int(res["num_gaussians"])
            return out.color[0, 0, 0, 0, 0]
        for p_ in enc.parameters():
            p_.grad = None
        res = encoder_forward(enc, dict(ctx), 0)
        out = dec(res["gaussians"][0], tgt_E, tgt_K, near_t, far_t, (H, W), depth_mode=None)
        loss = torch.nn.functional.mse_loss(out.color, target)
        loss.backward()
        info["gaussians"] = int(res["num_gaussians"])
        return loss

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if trace_steps:
        # a profiler is attached: nothing but plain steps between two markers the trace can be cut at
        torch.cuda.synchronize()
        mark = torch.full((64,), 0.25, device=dev)
        torch.erfinv(mark)                      # marker launch (a kernel name nothing else in the step uses: c3_step_glue.py)
        for _ in range(trace_steps):
            step()
        torch.erfinv(mark)
        torch.cuda.synchronize()
        return {"trace_steps": trace_steps, "gaussians": info["gaussians"]}
    # ---- whole step, events off ----
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    import time
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        loss = step()
        b.record()
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3 / steps
    each = [a.elapsed_time(b) for a, b in ev]
    step_ms = sorted(each)[len(each) // 2]          # median step (a step that has to grow the allocator's pool runs long)
    # ---- library kernels per stage (a second loop: every library launch bracketed by a HIP event pair) ----
    from freesplat_amd import rasterizer as _R
    streams, _R.NUM_STREAMS = _R.NUM_STREAMS, 1       # one stream: stage durations do not overlap and can be summed
    _lib.profile_collect()
    _lib.profile_enable(True)
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    _R.NUM_STREAMS = streams
    stages = {k: v[0] / steps for k, v in _lib.profile_collect().items() if v[0] > 0}
    lib_ms = sum(stages.values())
    other_ms = max(step_ms - lib_ms, 0.0)
    # The split of `other` into the stand-in modules' own kernels (out of scope) and the GLUE between the hot-path stages cannot be
    # taken from events (the stand-ins run interleaved with the glue, forward and backward): it comes from a rocprofv3 kernel
    # trace of this same step, every kernel classified by name (profiles/tools/c3_step_glue.py -> profiles/*_c3_step_glue.json).
    glue = committed_glue() if workload == "c3_train_step_hotpath" else None
    n_g = info["gaussians"]
    K_src = (num_views or V) - 1
    return {"metric": f"composed {'training' if train else 'evaluation'} steps/sec ({V} context views @ {H}x{W}, cost volume "
                      f"{H // 4}x{W // 4} K={K_src} D={D}, PTF fold, {n_targets} target views, {'fwd+bwd' if train else 'forward only'})",
            "value": 1e3 / step_ms, "unit": "steps/s", "ms_per_step": step_ms, "ms_each_step": [round(x, 3) for x in each], "wall_ms_per_step": wall_ms, "steps": steps,
            "dtype": "f32", "data": "synthetic (random images, seeded cameras; stand-in modules for the reference's out-of-scope networks)",
            "config": {"workload": workload, "image_hw": [H, W], "context_views": V, "target_views": n_targets,
                       "depth_planes": D, "match_hw": [H // 4, W // 4], "sources_per_view": K_src,
                       "raw_gaussians": V * H * W, "gaussians_after_fold": n_g, "sh_storage": "fp16" if sh_fp16 else "fp32"},
            "gaussians": n_g, "target_views": n_targets,
            "library_kernel_ms": lib_ms, "library_kernel_ms_by_stage": stages,
            "non_library_ms": other_ms,
            "non_library_note": "step - library kernels by HIP events on one stream: the stand-in modules' own kernels (convolutions, "
                                "the Linear layer: out of scope) + the torch / rocclr glue between the hot-path stages + GPU idle time",
            "glue_ms": None if glue is None else glue["glue_ms_per_step"],
            "glue_frac_of_gpu_time": None if glue is None else glue["glue_frac_of_hotpath_gpu_time"],
            "glue_source": None if glue is None else glue["source"],
            "glue_measured_in_this_run": False,
            "glue_note": "from the committed rocprofv3 kernel trace of this step, kernels classified by name: glue / (library kernels + "
                         "glue); the stand-ins' kernels are left out of the denominator",
            "loss": float(loss.detach()) if train else None}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--trace-steps", type=int, default=0,
                    help="run ONLY this many plain steps after the warm-up and print their count (for rocprofv3 --kernel-trace: "
                         "profiles/tools/c3_step_glue.py divides the trace by it)")
    ap.add_argument("--c4", action="store_true", help="config 4's evaluation step instead: 10 views at 384x512, K = 8, 8 targets, no autograd")
    ap.add_argument("--c5", action="store_true", help="config 5's evaluation step: 30 views at 384x512, K = 8, the 30-view fold, 8 targets, fp16 SH, no autograd")
    ap.add_argument("--small", action="store_true", help="config 1's size (256x256, 2 views, D = 16): a quick functional run")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    kw = dict(H=256, W=256, V=2, D=16, n_targets=2) if a.small else {}
    if a.c4:
        kw = dict(H=384, W=512, V=10, n_targets=8, train=False, num_views=9, workload="c4_eval_step_hotpath")
    if a.c5:
        kw = dict(H=384, W=512, V=30, n_targets=8, train=False, num_views=9, workload="c5_eval_step_hotpath", sh_fp16=True)
    if a.trace_steps:
        kw["trace_steps"] = a.trace_steps
    print(json.dumps(bench_c3_step(dev, a.steps, a.warmup, **kw)))
