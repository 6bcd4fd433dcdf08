"""Encoder -> cost-volume glue: host-side torch plumbing that prepares the 8
keyword arguments of `AVGFeatureVolumeManager.forward` exactly as
/root/reference/src/model/encoder/encoder_freesplat.py:216-288 does: intrinsics scaled to the matching
resolution (:217-220), source-view selection -- all other views, or the `num_views` pose-nearest ones
when there are more context views than that (:234-248, pose distance = |dt| + rotation angle, :40-60)
--, relative poses src<-cur / cur<-src (:253-258), gathered source features (:260-264), 4x4 K and
inverse K (:266-273), near/far of the first view (:276-277).  Device-agnostic; no kernels needed.
"""
from __future__ import annotations

import torch
from torch import Tensor


def rotation_distance(rotations: Tensor) -> Tensor:
    """encoder_freesplat.py:40-48: pairwise rotation angle, rotations [1,V,3,3] -> [V,V]."""
    R_rel = torch.matmul(rotations.unsqueeze(2).transpose(-2, -1), rotations.unsqueeze(1))
    trace = torch.diagonal(R_rel, dim1=-2, dim2=-1).sum(-1)
    trace = torch.clamp(trace, -1, 3)
    return torch.acos((trace - 1) / 2).squeeze(0)


def calculate_distance_matrix(poses: Tensor) -> Tensor:
    """encoder_freesplat.py:50-60: poses [1,V,4,4] (c2w) -> [V,V] translation + rotation distance."""
    t = poses[:, :, :3, 3]
    return torch.cdist(t, t).squeeze(0) + rotation_distance(poses[:, :, :3, :3])


def select_source_views(extrinsics: Tensor, num_context_views: int) -> Tensor:
    """[b,V,4,4] -> src_indices [b,V,K] (ascending view ids per row).  K = V-1 when V <= num_context_views,
    else num_context_views-1 nearest views by pose distance (the top-k includes the view itself)."""
    b, V = extrinsics.shape[:2]
    dev = extrinsics.device
    cur = torch.arange(V, device=dev)
    full = cur[None].repeat(V, 1)
    not_self = ~(full == cur[:, None])
    if V <= num_context_views:
        return full[not_self].view(1, V, V - 1).repeat(b, 1, 1)
    k = min(num_context_views, V)
    slide = torch.zeros((V, V), dtype=torch.bool, device=dev)
    _, idx = torch.topk(calculate_distance_matrix(extrinsics), k, largest=False, sorted=False, dim=1)
    slide.scatter_(1, idx, True)
    slide[cur, cur] = False
    return full[not_self * slide].view(1, V, k - 1).repeat(b, 1, 1)


def prepare_cost_volume_inputs(extrinsics: Tensor, intrinsics: Tensor, matching_feats: Tensor, near: Tensor,
                               far: Tensor, image_hw: tuple[int, int], num_context_views: int,
                               rows: range | None = None) -> dict:
    """extrinsics [b,V,4,4] c2w, intrinsics [b,V,3,3] normalised, matching_feats [(b V),C,h/4,w/4] (level-1
    backbone features), near/far [b,V].  Returns the kwargs of cost_volume.forward (B = b*V rows).

    The reference repeats every tensor V-fold and `gather`s the source rows out of the copies
    (encoder_freesplat.py:255-264: V*V*C*h/4*w/4 floats for the features); here the source rows are indexed
    directly -- same values, no V-fold intermediate.  `rows` (a range of current views, b = 1): only those rows of the
    B dimension are materialised -- what a view-sharded caller needs (cost_volume.sharded_cost_volume)."""
    b, V = extrinsics.shape[:2]
    h, w = image_hw
    dev = extrinsics.device
    K = intrinsics.clone()
    K[:, :, 0] *= (w // 4)
    K[:, :, 1] *= (h // 4)
    src_indices = select_source_views(extrinsics, num_context_views)                 # [b,V,Ks]
    cur = slice(None)
    if rows is not None:
        if b != 1:
            raise NotImplementedError("rows=...: one scene per call (b = 1)")
        cur = slice(rows.start, rows.stop)
        src_indices = src_indices[:, cur]
    Vr, Ks = src_indices.shape[1:]
    bi = torch.arange(b, device=dev)[:, None, None]
    pick = lambda t: t[bi, src_indices]                                             # [b,V,...] -> [b,Vr,Ks,...]
    src_extr = pick(extrinsics)                                                      # [b,Vr,Ks,4,4]
    src_K3 = pick(K)
    inv = lambda t: torch.linalg.inv_ex(t).inverse
    cur_extr = extrinsics[:, cur]
    src_cam_T_cur = inv(src_extr) @ cur_extr.unsqueeze(2)
    cur_cam_T_src = inv(cur_extr).unsqueeze(2) @ src_extr
    C, h4, w4 = matching_feats.shape[-3:]
    feats = matching_feats.view(b, V, C, h4, w4)
    src_feats = pick(feats).reshape(b * Vr, Ks, C, h4, w4)
    src_K = torch.eye(4, device=dev)[None, None].repeat(b * Vr, Ks, 1, 1)
    src_K[:, :, :3, :3] = src_K3.reshape(b * Vr, Ks, 3, 3)
    cur_inv = torch.eye(4, device=dev)[None].repeat(b * Vr, 1, 1)
    cur_inv[:, :3, :3] = inv(K[:, cur].reshape(b * Vr, 3, 3))
    return dict(cur_feats=feats[:, cur].reshape(b * Vr, C, h4, w4), src_feats=src_feats,
                src_extrinsics=src_cam_T_cur.reshape(b * Vr, Ks, 4, 4), src_poses=cur_cam_T_src.reshape(b * Vr, Ks, 4, 4),
                src_Ks=src_K, cur_invK=cur_inv, min_depth=near[:1, 0].type_as(src_K).view(1, 1, 1, 1),
                max_depth=far[:1, 0].type_as(src_K).view(1, 1, 1, 1))
