"""CPU restatement (torch, vectorised over the D depth planes) of FreeSplat's plane-sweep cost volume.

TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this.  Pinned against the reference itself: tests/golden/cv_small_k{1,2}.npz and
cv_native_stat.json were produced by importing /root/reference (tests/golden/make_golden.py) and
tests/test_cost_volume_oracle.py checks this file against them.

Follows (paths under /root/reference):
  src/model/encoder/modules/cost_volume.py:98-134   generate_depth_planes
  src/model/encoder/modules/cost_volume.py:429-619  AVGFeatureVolumeManager.build_cost_volume
  sr_utils/geometry_utils.py:22-59, 62-89           BackprojectDepth, Project3D
  src/model/encoder/modules/networks.py:218-236     MLP (Linear-LeakyReLU(0.01)-Linear-LeakyReLU-Linear)
in the closed form of SURVEY.md Appendix B: a plane-induced homography per (pixel, source).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import Tensor


def depth_planes(min_depth, max_depth, D: int) -> Tensor:
    """cost_volume.py:116-125: uniform in inverse depth, plane 0 = min_depth."""
    ramp = torch.linspace(0, 1, D)
    inv_min, inv_max = 1.0 / float(min_depth), 1.0 / float(max_depth)
    return 1.0 / (torch.tensor(inv_min) + (torch.tensor(inv_max) - torch.tensor(inv_min)) * ramp)


def cost_volume(cur_feats: Tensor, src_feats: Tensor, src_extrinsics: Tensor, src_Ks: Tensor,
                cur_invK: Tensor, min_depth, max_depth, D: int, mlp: dict, return_pre: bool = False):
    """cur_feats [B,C,h,w], src_feats [B,K,C,h,w], src_extrinsics [B,K,4,4] (src<-cur),
    src_Ks [B,K,4,4], cur_invK [B,4,4]; mlp = {"w1" [32,C+1], "b1", "w2" [32,32], "b2", "w3" [1,32], "b3"}.
    Returns [B,D,h,w]  (with return_pre: also the two hidden layers' pre-activations [B,D,h*w,32] and the sample positions --
    the gradient tests mask the points that sit on a LeakyReLU kink or on the border of a source image).  Runs in the dtype of `cur_feats` (float64 for gradient references;
    the plane depths are the fp32 values the module generates, cast)."""
    B, K, C, h, w = src_feats.shape
    dt = cur_feats.dtype
    planes = depth_planes(min_depth.reshape(-1)[0] if torch.is_tensor(min_depth) else min_depth,
                          max_depth.reshape(-1)[0] if torch.is_tensor(max_depth) else max_depth, D).to(dt)
    ys, xs = torch.meshgrid(torch.arange(h, dtype=dt), torch.arange(w, dtype=dt), indexing="ij")
    pix = torch.stack([xs + 0.5, ys + 0.5, torch.ones_like(xs)], 0).reshape(3, -1)       # geometry_utils.py:33-44
    rays = cur_invK[:, :3, :3] @ pix                                                       # [B,3,N]   :56
    pts = planes.view(1, D, 1, 1) * rays[:, None]                                          # [B,D,3,N] :57
    P = (src_Ks @ src_extrinsics)[:, :, :3]                                                # [B,K,3,4] :78-80
    cam = torch.einsum("bkij,bdjn->bkdin", P[..., :3], pts) + P[..., 3][:, :, None, :, None]  # [B,K,D,3,N]
    z = cam[:, :, :, 2]
    mask = z.abs() > 1e-8                                                                  # :83
    depth = z + 1e-8                                                                       # :84
    scale = torch.where(mask, 1.0 / depth, torch.ones_like(depth))                         # :85
    px, py = cam[:, :, :, 0] * scale, cam[:, :, :, 1] * scale                              # :87
    # cost_volume.py:536-549: uv = 2*pix/(w,h) - 1, bilinear grid_sample, zeros padding, align_corners=False
    grid = torch.stack([2 * px / w - 1, 2 * py / h - 1], -1).reshape(B * K, D, h * w, 2)
    warped = F.grid_sample(src_feats.reshape(B * K, C, h, w), grid, mode="bilinear", padding_mode="zeros",
                           align_corners=False).reshape(B, K, C, D, h * w)
    m = (depth > 0).to(warped.dtype)                                                       # :571-572
    dot = (warped * cur_feats.reshape(B, 1, C, 1, h * w)).sum(2) * m                       # [B,K,D,N] :589-593
    valid = (dot != 0)
    cnt = valid.sum(1, keepdim=True) + 1e-8                                                # :595
    dot_mean = dot.sum(1, keepdim=True) / cnt                                              # [B,1,D,N]
    feat_mean = (warped * valid.unsqueeze(2)).sum(1) / cnt                                 # [B,C,D,N]   :597-598
    x = torch.cat([feat_mean, dot_mean], 1).permute(0, 2, 3, 1)                            # [B,D,N,C+1] :600-607
    z1 = F.linear(x, mlp["w1"], mlp["b1"])
    z2 = F.linear(F.leaky_relu(z1, 0.01), mlp["w2"], mlp["b2"])
    x = F.linear(F.leaky_relu(z2, 0.01), mlp["w3"], mlp["b3"])                             # networks.py:218-236
    if return_pre:   # + the sample positions in source texel coordinates [B,K,D,N] (taps at floor, floor + 1)
        return x.reshape(B, D, h, w), dict(z1=z1, z2=z2, ix=px - 0.5, iy=py - 0.5, feat_mean=feat_mean, dot_mean=dot_mean,
                                           cnt=cnt, valid=valid, front=depth > 0, P=P, planes=planes)
    return x.reshape(B, D, h, w)


def mlp_from_state(sd: dict, prefix: str = "mlp__net__") -> dict:
    g = lambda k: torch.as_tensor(sd[prefix + k]).float()
    return dict(w1=g("0__weight"), b1=g("0__bias"), w2=g("2__weight"), b2=g("2__bias"), w3=g("4__weight"),
                b3=g("4__bias"))
