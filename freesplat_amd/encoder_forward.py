"""`EncoderFreeSplat.forward` for the drop-in: the reference's own sub-modules, called in the
reference's order, with the hot-path glue between them replaced.

    compat.patch_reference() binds   src.model.encoder.encoder_freesplat.EncoderFreeSplat.forward = encoder_forward

What is NOT ours stays a module call on `self` (backbone, cv_encoder, depth_decoder, high_resolution_skip,
to_gaussians, weight_embedding: out of scope).  What changes against
/root/reference/src/model/encoder/encoder_freesplat.py:190-429:
  * :216-288  the V-fold `repeat` + `gather` of extrinsics / intrinsics / IMAGES / matching features that prepares the
              cost-volume call -> encoder_glue.prepare_cost_volume_inputs (source rows indexed directly; the gathered
              source images, `src_image`, are never used by the reference and are not formed);
  * :280-288  self.cost_volume(...)                         (HIP sweep once patch_reference has rebound the class)
  * :317-326  self.gaussian_adapter.forward(fusion=True)    (fs_unproject_*: no b x V python loop)
  * :364-368  self.fuse_gaussians(...)                      (fs_ptf_fold)
  * :371-386  to_gaussians + gaussian_adapter.forward(fusion=False, coords=...)   (fs_gaussian_head_*)
Same signature and the same result dictionary (keys and shapes), so model_wrapper.py:231-233,292,315-322,383,425 and
the losses read it unchanged.  tests/test_compat_reference.py runs this function and the reference's forward on the SAME
reference modules (CPU) and compares every entry of the two dictionaries.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor, nn

from .encoder_glue import prepare_cost_volume_inputs


def _pixel_grid(h: int, w: int, device) -> Tensor:
    """Pixel-centre coordinates in (0, 1), (x, y) order, [h*w, 1, 2] (src/geometry/projection.py:117-137)."""
    ys = (torch.arange(h, device=device) + 0.5) / h
    xs = (torch.arange(w, device=device) + 0.5) / w
    gx, gy = torch.meshgrid(xs, ys, indexing="xy")
    return torch.stack([gx, gy], dim=-1).reshape(h * w, 1, 2)


def _bv(t: Tensor, b: int) -> Tensor:
    """[(b v), c, h, w] -> [b, v, (c h w), 1, 1]"""
    return t.reshape(b, t.shape[0] // b, -1, 1, 1)


def encoder_forward(self, context, global_step: int, deterministic: bool = False,
                    visualization_dump: Optional[dict] = None, is_testing: bool = False, export_ply: bool = False,
                    dataset_name: str = "scannet") -> dict:
    images = context["image"]
    device = images.device
    b, V, _, h, w = images.shape
    context["image_shape"] = (h, w)
    results: dict = {}

    # backbone per scene (its batch-norm layers are put in train() mode, as encoder_freesplat.py:78-80,208 does)
    for m in self.backbone.modules():
        if isinstance(m, nn.modules.batchnorm._BatchNorm):
            m.train()
    per_scene = [self.backbone(images[i]) for i in range(b)]
    feats = [torch.cat([p[level] for p in per_scene], dim=0) for level in range(len(per_scene[0]))]

    # cost volume on the 1/4-resolution level, its encoder and the depth decoder
    volume = self.cost_volume(**prepare_cost_volume_inputs(context["extrinsics"], context["intrinsics"], feats[1],
                                                           context["near"], context["far"], (h, w), self.cfg.num_views))
    feats = feats[:1] + self.cv_encoder(volume, feats[1:])
    dec = self.depth_decoder(feats)

    # per-pixel latents (+ full-resolution skip), densities, depths, depth weights
    flat_images = images.reshape(b * V, *images.shape[2:])
    head = dec["output_pred_s-1_b1hw"]
    skip = self.high_resolution_skip[0](flat_images)
    if head.is_cuda and head.shape[1] == 65 and skip.shape[1] == 64:
        # the fold reads pixel-major rows: head[:, 1:] + skip and the channel-major -> pixel-major move in one HIP pass each way
        # (gaussian_adapter.latents_pack) instead of an add and three transposing 1 GB copies per training step at config 3
        from .gaussian_adapter import latents_pack
        lat, dens_raw = latents_pack(head, skip)
        latents = lat.reshape(b, V, h * w, -1)
        densities = torch.sigmoid(dens_raw).reshape(b, V, h * w, 1, 1)
    else:       # host tensors (tests/test_compat_reference.py runs this glue on the reference's own CPU modules)
        latents = (head[:, 1:] + skip).reshape(b, V, -1, h * w).transpose(-1, -2)
        densities = torch.sigmoid(_bv(head[:, :1], b))
    depths = _bv(dec["depth_pred_s-1_b1hw"], b)
    weights = _bv(dec["depth_weights"], b)
    xy_ray = _pixel_grid(h, w, device) + torch.zeros(b, V, h * w, self.cfg.num_surfaces, 2, device=latents.device)

    unproj = lambda t, n: t.reshape(*t.shape[:2], 1, 1, 1, n, n)
    coords = self.gaussian_adapter.forward(unproj(context["extrinsics"], 4), unproj(context["intrinsics"], 3),
                                           xy_ray[:, :, :, :, None], depths, densities, latents, (h, w), fusion=True)

    results["depth_num0_s-1"] = depths
    _record_gt_depth(results, context, "depth_s-1", "depth_num0_s-1_raw", "depth_num0_s-1_mask", per_pixel=True)
    results["depth_num0_s-1_b1hw"] = dec["depth_pred_s-1_b1hw"]
    for s in range(self.max_depth):
        results[f"depth_num0_s{s}"] = _bv(dec[f"depth_pred_s{s}_b1hw"], b)
        _record_gt_depth(results, context, f"depth_s{s}", f"depth_num0_s{s}_raw_b1hw", f"depth_num0_s{s}_mask_b1hw")
        results[f"depth_num0_s{s}_b1hw"] = dec[f"depth_pred_s{s}_b1hw"]

    # Pixel-wise Triplet Fusion and the Gaussian head, scene by scene
    n_raw = V * h * w
    depth_maps = dec["depth_pred_s-1_b1hw"].reshape(b, V, *dec["depth_pred_s-1_b1hw"].shape[1:])
    fused = []
    # (b = 1, the only batch size the reference's configurations use: the whole tensors, not `x[0:1]` -- a slice's backward
    #  zero-fills a full-size gradient and copies into it: two 1 GB fills + copies per config-3 training step for the latents)
    whole = (lambda t: t) if b == 1 else None
    for i in range(b):
        one = slice(i, i + 1)
        pick = whole if whole is not None else (lambda t: t[one])
        lat, xyz, extr, dep = self.fuse_gaussians([pick(latents)], [pick(coords)], pick(densities), pick(weights), depth_maps[i],
                                                  pick(context["extrinsics"]), pick(context["intrinsics"]), (h, w))
        raw = self.to_gaussians(lat)
        raw = raw.reshape(*raw.shape[:-1], self.cfg.num_surfaces, -1)                    # [1, M, srf, 2 + d_in]
        M = raw.shape[1]
        fused.append(self.gaussian_adapter.forward(
            extr[:, None, :, None, None],                                                # [1, 1, M, 1, 1, 4, 4]
            context["intrinsics"][one, 0][:, None, None, None, None].expand(1, 1, M, 1, 1, 3, 3),
            xy_ray[one, :, :, :, None],
            dep[:, None, :, None, None],
            torch.sigmoid(raw[..., :1])[:, None],                                        # [1, 1, M, srf, 1]
            raw[..., 2:][:, None, :, :, None, :],                                        # [1, 1, M, srf, 1, d_in]
            (h, w), fusion=False, coords=xyz[:, None, :, None, None, :]))
    last = fused[-1]
    n_out = fused[0].means.shape[2]
    results["gs_ratio"] = n_out / n_raw
    results["num_gaussians"] = n_out
    flat = lambda t, k: t.reshape(t.shape[0], -1, *t.shape[-k:]) if k else t.reshape(t.shape[0], -1)
    results["visualizations"] = {"scales": flat(last.scales, 1) if last.scales.dim() == 6 else last.scales,
                                 "rotations": flat(last.rotations, 1) if last.rotations.dim() == 6 else last.rotations}
    Gaussians = _gaussians_type(self)
    results["gaussians"] = [Gaussians(flat(g.means, 1), flat(g.covariances, 2), flat(g.harmonics, 2), flat(g.opacities, 0))
                            for g in fused]
    return results


def _record_gt_depth(results: dict, context, key: str, raw_key: str, mask_key: str, per_pixel: bool = False) -> None:
    """Ground-truth depth of a scale, when the batch carries it (encoder_freesplat.py:331-336, 345-351)."""
    gt = context.get(key) if hasattr(context, "get") else None
    if gt is None:
        return
    b, V = gt.shape[:2]
    raw = gt.reshape(b, V, gt.shape[2], -1).transpose(-1, -2)[..., None] if per_pixel else gt.reshape(b * V, *gt.shape[2:])
    results[raw_key] = raw
    results[mask_key] = (raw > 1e-3) * (raw < 10)


def _gaussians_type(encoder):
    """The `Gaussians` dataclass of the tree the encoder lives in (src/model/types.py:7-12) -- the reference's own when
    patched into it, ours otherwise."""
    import importlib
    import sys
    mod = sys.modules.get("src.model.types")
    if mod is None:
        try:
            mod = importlib.import_module("src.model.types")
        except Exception:
            from .decoder import Gaussians
            return Gaussians
    return mod.Gaussians
