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
tests/test_composed_dropin.py, so that every hot-path stage sees config 3's tensor sizes.  Their own time is measured apart
(the same modules run alone, forward + backward, on the captured inputs) and is not the subject.

Reported (one stream, HIP events): ms per whole step; ms of library kernels per stage (fs_profile_* hooks: cost_volume,
encoder_tail, ptf, preprocess, tile_scan, render, render_bwd, preprocess_bwd); ms of the stand-in modules alone; and the
REST = step - library - stand-ins: the torch / rocclr glue between the stages (reshapes that copy, index kernels, fills,
reductions, the loss, host-induced gaps), i.e. what the inter-stage plumbing costs on the GPU's clock.
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
        self.c0 = nn.Conv2d(3, 8, 3, stride=2, padding=1)
        self.c1 = nn.Conv2d(8, C, 3, stride=2, padding=1)

    def forward(self, x):
        f0 = torch.tanh(self.c0(x))
        return [f0, self.c1(f0)]


class _CVEncoder(nn.Module):
    def __init__(self, D, C):
        super().__init__()
        self.c = nn.Conv2d(D + C, 24, 3, padding=1)

    def forward(self, volume, feats):
        return [torch.tanh(self.c(torch.cat([volume, feats[0]], 1)))]


class _DepthTrunk(nn.Module):
    """The depth decoder's convolutions (stand-in): plane logits at half resolution, 1 + 64 head channels at full."""

    def __init__(self, D):
        super().__init__()
        self.conv_depth = nn.Conv2d(8 + 24, D, 3, padding=1)
        self.conv_last = nn.Conv2d(8 + 24, 1 + 64, 3, padding=1)

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

    def __init__(self, H, W, V, D, C):
        super().__init__()
        from freesplat_amd.cost_volume import AVGFeatureVolumeManager
        from freesplat_amd.gaussian_adapter import GaussianAdapter, GaussianAdapterCfg
        from freesplat_amd.ptf import GRU
        self.cfg = types.SimpleNamespace(num_views=V, num_surfaces=1)
        self.max_depth = 1
        torch.manual_seed(11)
        self.backbone = _Backbone(C)
        self.cv_encoder = _CVEncoder(D, C)
        self.high_resolution_skip = nn.ModuleList([nn.Conv2d(3, 64, 3, padding=1)])
        self.to_gaussians = nn.Sequential(nn.ReLU(), nn.Linear(64, 36))
        self.cost_volume = AVGFeatureVolumeManager(H // 4, W // 4, num_depth_bins=D, mlp_channels=[202, 32, 32, 1],
                                                   matching_dim_size=C)
        self.gru = GRU()
        self.depth_decoder = _DepthDecoder(D)
        self.gaussian_adapter = GaussianAdapter(GaussianAdapterCfg(0.5, 15.0, 2))

    def fuse_gaussians(self, *a, **k):
        from freesplat_amd.ptf import fuse_gaussians
        return fuse_gaussians(self, *a, **k)


def _standins(enc):
    return {"backbone": enc.backbone, "cv_encoder": enc.cv_encoder, "depth_trunk": enc.depth_decoder.trunk,
            "high_resolution_skip": enc.high_resolution_skip[0], "to_gaussians": enc.to_gaussians}


def _flat_tensors(o):
    if torch.is_tensor(o):
        return [o]
    if isinstance(o, (list, tuple)):
        return [t for x in o for t in _flat_tensors(x)]
    if isinstance(o, dict):
        return [t for x in o.values() for t in _flat_tensors(x)]
    return []


def _snapshot(o):
    if torch.is_tensor(o):
        return ("t", o.detach(), bool(o.requires_grad))
    if isinstance(o, (list, tuple)):
        return ("l", [_snapshot(x) for x in o])
    return ("o", o)


def _restore(s):
    if s[0] == "t":
        return s[1].clone().requires_grad_(s[2])
    if s[0] == "l":
        return [_restore(x) for x in s[1]]
    return s[1]


def bench_c3_step(dev, steps=3, warmup=2, H=968, W=1296, V=3, D=128, C=48, n_targets=4) -> dict:
    import inputs
    from freesplat_amd import _lib
    from freesplat_amd.decoder import DecoderSplattingCUDA
    from freesplat_amd.encoder_forward import encoder_forward
    enc = _Encoder(H, W, V, D, C).to(dev)
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
        for p_ in enc.parameters():
            p_.grad = None
        res = encoder_forward(enc, dict(ctx), 0)
        out = dec(res["gaussians"][0], tgt_E, tgt_K, near_t, far_t, (H, W), depth_mode=None)
        loss = ((out.color - target) ** 2).mean()
        loss.backward()
        info["gaussians"] = int(res["num_gaussians"])
        return loss

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
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
    step_ms = sum(a.elapsed_time(b) for a, b in ev) / steps
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
    # ---- the stand-in modules alone (forward + backward on the inputs they saw in a step) ----
    captured = {}
    hooks = [m.register_forward_hook(lambda mod, args, out, n=n: captured.__setitem__(n, _snapshot(args)),
                                     with_kwargs=False) for n, m in _standins(enc).items()]
    step()
    for h_ in hooks:
        h_.remove()
    standin = {}
    for n, m in _standins(enc).items():
        args = _restore(captured[n])        # (leaves that require grad exactly where the step's inputs did)

        def run():
            outs = _flat_tensors(m(*args))
            sum(o.sum() for o in outs).backward()
        run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(steps):
            run()
        b.record()
        torch.cuda.synchronize()
        standin[n] = a.elapsed_time(b) / steps
    standin_ms = sum(standin.values())
    glue_ms = max(step_ms - lib_ms - standin_ms, 0.0)
    hot_ms = lib_ms + glue_ms                           # the hot path's own GPU time: its kernels + its plumbing
    n_g = info["gaussians"]
    return {"metric": f"composed config-3 training steps/sec ({V} context views @ {H}x{W}, cost volume {H // 4}x{W // 4} K={V - 1} "
                      f"D={D}, PTF fold, {n_targets} target views, fwd+bwd)",
            "value": 1e3 / step_ms, "unit": "steps/s", "ms_per_step": step_ms, "wall_ms_per_step": wall_ms, "steps": steps,
            "dtype": "f32", "data": "synthetic (random images, seeded cameras; stand-in modules for the reference's out-of-scope networks)",
            "config": {"workload": "c3_train_step_hotpath", "image_hw": [H, W], "context_views": V, "target_views": n_targets,
                       "depth_planes": D, "match_hw": [H // 4, W // 4], "sources_per_view": V - 1,
                       "raw_gaussians": V * H * W, "gaussians_after_fold": n_g},
            "gaussians": n_g, "target_views": n_targets,
            "library_kernel_ms": lib_ms, "library_kernel_ms_by_stage": stages,
            "standin_modules_ms": standin_ms, "standin_modules_ms_by_module": standin,
            "glue_ms": glue_ms, "glue_frac_of_gpu_time": glue_ms / max(hot_ms, 1e-9),
            "glue_frac_note": "glue / (library kernels + glue): the stand-ins' own convolutions are out of scope and left out of "
                              "the denominator; glue = step - library - stand-ins by HIP events on one stream, so it also holds "
                              "any GPU idle time the host causes",
            "loss": float(loss.detach())}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--small", action="store_true", help="config 1's size (256x256, 2 views, D = 16): a quick functional run")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    kw = dict(H=256, W=256, V=2, D=16, n_targets=2) if a.small else {}
    print(json.dumps(bench_c3_step(dev, a.steps, a.warmup, **kw)))
