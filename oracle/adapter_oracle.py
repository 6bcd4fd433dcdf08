"""CPU restatement (torch) of FreeSplat's GaussianAdapter on the fused path.

TEST INFRASTRUCTURE ONLY (see oracle/raster_oracle.c header for the rule).  Pinned against the
reference itself: tests/golden/adapter_small.npz (fusion=False) and the `coords` array of
tests/golden/ptf_small.npz (fusion=True), both produced by importing /root/reference
(tests/golden/make_golden.py); tests/test_adapter_oracle.py checks this file against them.

Follows /root/reference/src/model/encoder/common/gaussian_adapter.py:19-95 (Create_from_depth_map),
:135-214 (GaussianAdapter.forward, get_scale_multiplier) and common/gaussians.py:8-44
(quaternion_to_matrix xyzw, build_covariance).
"""
from __future__ import annotations

import torch
from torch import Tensor


def unproject(depths: Tensor, extrinsics: Tensor, k0_pix: Tensor, h: int, w: int) -> Tensor:
    """fusion=True (gaussian_adapter.py:174-188, 36-79): depths [V,h*w], extrinsics c2w [V,4,4], k0_pix =
    (fx, fy, cx, cy) of VIEW 0 in pixels (the reference builds Create_from_depth_map once from
    intrinsics[i,0], :177-181).  Integer pixel coordinates, no +0.5.  Returns world xyz [V,h*w,3]."""
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    part = torch.stack([(xs - k0_pix[2]) / k0_pix[0], (ys - k0_pix[3]) / k0_pix[1]], -1).reshape(-1, 2)   # :45
    out = []
    for v in range(depths.shape[0]):
        z = depths[v].reshape(-1, 1)
        xyz1 = torch.cat([part * z, z, torch.ones_like(z)], -1)                                            # :62-64
        out.append((extrinsics[v] @ xyz1.T)[:3].T)                                                         # :68-70
    return torch.stack(out)


def quaternion_to_matrix(q: Tensor, eps: float = 1e-8) -> Tensor:
    i, j, k, r = q.unbind(-1)                                                                              # gaussians.py:15 (xyzw)
    two_s = 2 / ((q * q).sum(-1) + eps)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def gaussian_head(raw: Tensor, depths: Tensor, extrinsics: Tensor, multiplier: Tensor, sh_mask: Tensor,
                  scale_min: float = 0.5, scale_max: float = 15.0, eps: float = 1e-8):
    """fusion=False with coords given (gaussian_adapter.py:151-172, 191-201).  raw [M,34] = (scales 3,
    rotation xyzw 4, sh 27 as (xyz, d_sh)), depths [M], extrinsics [M,4,4] (blended, generally not
    rigid), multiplier scalar/[M] = 0.1 * sum(K[:2,:2]^-1 @ (1/w, 1/h)) (:203-214).
    Returns covariances [M,3,3], harmonics [M,3,9], scales [M,3], rotations [M,4]."""
    s_raw, r_raw, sh_raw = raw.split((3, 4, 27), dim=-1)
    scales = scale_min + (scale_max - scale_min) * s_raw.sigmoid()                                         # :155-157
    scales = scales * depths[..., None] * (multiplier[..., None] if multiplier.dim() else multiplier)     # :160
    rot = r_raw / (r_raw.norm(dim=-1, keepdim=True) + eps)                                                 # :163
    sh = sh_raw.reshape(-1, 3, 9) * sh_mask                                                                # :166-167
    R = quaternion_to_matrix(rot)
    S = scales.diag_embed()
    cov = R @ S @ S.transpose(-1, -2) @ R.transpose(-1, -2)                                                # gaussians.py:38-44
    Rc = extrinsics[..., :3, :3]
    cov = Rc @ cov @ Rc.transpose(-1, -2)                                                                  # :171-172
    return cov, sh, scales, rot


def scale_multiplier(intrinsics: Tensor, h: int, w: int, multiplier: float = 0.1) -> Tensor:
    """get_scale_multiplier (:203-214) for normalised intrinsics [...,3,3]."""
    px = 1 / torch.tensor((w, h), dtype=torch.float32)
    return (multiplier * (intrinsics[..., :2, :2].inverse() @ px)).sum(-1)
