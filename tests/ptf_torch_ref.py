"""Test-only cross-check of the PTF fold: the reference's update block (encoder_freesplat.py:485-519) op by op in torch
device ops on the HIP index lists (fs_ptf_match), differentiable through autograd, one host sync per view.  The product
(freesplat_amd.ptf.fuse_gaussians) runs the fold and its backward on the HIP kernels; this formulation only checks them."""
import torch

from freesplat_amd.ptf import match_view, world_to_camera
from oracle.ptf_oracle import gru as oracle_gru, positional_encoding     # the reference-pinned restatements


def fuse_gaussians_torch(self, gaussians, coords, densities, weight_emb, depths, extrinsics, intrinsics, image_shape,
                         depth_thres=0.1):
    """The fold step by step in torch device ops on the HIP index lists (fs_ptf_match), differentiable through
    autograd; one host sync per view.  The cross-check of _PtfFold's backward (the product folds one scene per call, like the
    reference's caller, and raises for b > 1)."""
    length = gaussians[0].shape[1]
    G = gaussians[0][:, 0]
    R = densities[:, 0]
    O = weight_emb[:, 0]
    X = coords[0][:, 0, :, 0, 0]
    Ex = extrinsics[:, 0][:, None].repeat(1, G.shape[1], 1, 1)
    depths = depths.reshape(depths.shape[0], -1)
    Dp = depths[None, 0]
    h, w = image_shape
    for i in range(1, length):
        extrinsic = extrinsics[0, i]
        K = intrinsics[0, i].clone()
        K[:1, :] *= w
        K[1:2, :] *= h
        kpix = torch.stack([K[0, 0], K[1, 1], K[0, 2], K[1, 2]])
        w2c = world_to_camera(extrinsic[None]).view(4, 4)      # (the product's own matrices: ptf.world_to_camera)
        keep, fuse, fpix, app = match_view(X[0], w2c, kpix, depths[i], h, w, depth_thres)
        if fuse.numel() > 0:
            xe = positional_encoding(torch.cat([R[:, fuse], weight_emb[:, i, fpix]], dim=-1), 6)
            he = positional_encoding(torch.cat([densities[:, i, fpix], O[:, fuse]], dim=-1), 6)
            n_f = fuse.numel()      # the oracle's GRU (oracle/ptf_oracle.py:gru) on the module's parameters: autograd reaches them
            fused = oracle_gru(dict(self.gru.named_parameters()), gaussians[0][0, i, fpix], G[0, fuse], xe.reshape(n_f, 24),
                               he.reshape(n_f, 24))[None]
            w0 = R[:, fuse].repeat(1, 1, 1, 2)
            w1 = densities[:, i, fpix].repeat(1, 1, 1, 2)
            G = torch.cat([G[:, keep], fused], dim=1)
            X = torch.cat([X[:, keep], (X[:, fuse] * w0[..., 1] + coords[0][:, i, fpix, 0, 0] * w1[..., 1])
                           / (w0[..., 1] + w1[..., 1])], dim=1)
            Ex = torch.cat([Ex[:, keep], (Ex[:, fuse] * w0[..., :1] + extrinsics[:, i, None] * w1[..., :1])
                            / (w0[..., :1] + w1[..., :1])], dim=1)
            Dp = torch.cat([Dp[:, keep], (Dp[:, fuse] * w0[..., 0, 0] + depths[None, i, fpix] * w1[..., 0, 0])
                            / (w0[..., 0, 0] + w1[..., 0, 0])], dim=1)
            R_new = R[:, fuse] + densities[:, i, fpix]
            O_new = O[:, fuse] + weight_emb[:, i, fpix]
            R = torch.cat([R[:, keep], R_new], dim=1)
            O = torch.cat([O[:, keep], O_new], dim=1)
        G = torch.cat([G, gaussians[0][:, i, app]], dim=1)
        X = torch.cat([X, coords[0][:, i, app, 0, 0]], dim=1)
        R = torch.cat([R, densities[:, i, app]], dim=1)
        O = torch.cat([O, weight_emb[:, i, app]], dim=1)
        Ex = torch.cat([Ex, extrinsics[:, i, None].repeat(1, app.numel(), 1, 1)], dim=1)
        Dp = torch.cat([Dp, depths[None, i, app]], dim=1)
    return G, X, Ex, Dp
