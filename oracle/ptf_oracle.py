"""CPU restatement of FreeSplat's Pixel-wise Triplet Fusion fold (torch + numpy, CPU).

TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this.  Pinned against the reference itself: tests/golden/ptf_small.npz and ptf_tie.npz were
produced by importing /root/reference and calling EncoderFreeSplat.fuse_gaussians
(tests/golden/make_golden.py); tests/test_ptf_oracle.py checks this file against them, including
the output ORDER.

Follows /root/reference/src/model/encoder/encoder_freesplat.py:431-522 (fuse_gaussians), :62-77
(positional_encoding) and src/model/encoder/modules/networks.py:188-214 (GRU) in the form of
SURVEY.md Appendix C.  The index-producing part (`match_step`) is written with explicit fp32
elementwise arithmetic in a fixed order so that the HIP kernels can be bit-exact against it.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor

f32 = np.float32


def match_step(xyz: np.ndarray, w2c: np.ndarray, kpix: np.ndarray, depth_i: np.ndarray, h: int, w: int,
               depth_thres: float = 0.1):
    """One view's matching.  xyz [M,3], w2c [4,4] = inverse(extrinsics_i), kpix = (fx, fy, cx, cy) in
    pixels, depth_i [h*w].  Returns (keep_idx, fuse_idx, fuse_pix, append_pix) ascending int64 arrays.
    encoder_freesplat.py:454-482, 508."""
    xyz = np.asarray(xyz, f32); w2c = np.asarray(w2c, f32); kpix = np.asarray(kpix, f32)
    depth_i = np.asarray(depth_i, f32).reshape(-1)
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    with np.errstate(all="ignore"):
        cx = ((w2c[0, 0] * x + w2c[0, 1] * y) + w2c[0, 2] * z) + w2c[0, 3]
        cy = ((w2c[1, 0] * x + w2c[1, 1] * y) + w2c[1, 2] * z) + w2c[1, 3]
        cz = ((w2c[2, 0] * x + w2c[2, 1] * y) + w2c[2, 2] * z) + w2c[2, 3]
        px = (cx / cz) * kpix[0] + kpix[2]                                    # :457-459
        py = (cy / cz) * kpix[1] + kpix[3]
        col, row = np.rint(px), np.rint(py)                                   # :460 round half to even
        valid = (row >= 0) & (row < h) & (col >= 0) & (col < w) & (cz > 0)    # :461
    P = h * w
    pix = np.full(xyz.shape[0], -1, np.int64)
    pix[valid] = row[valid].astype(np.int64) * w + col[valid].astype(np.int64)
    zbuf = np.full(P, 10000.0, f32)                                           # :464
    np.minimum.at(zbuf, pix[valid], cz[valid])                                # :466 scatter amin
    fm = np.abs(zbuf - depth_i) < np.maximum(depth_i * f32(0.05), f32(depth_thres))   # :468
    win = np.zeros(xyz.shape[0], bool)
    win[valid] = (zbuf[pix[valid]] == cz[valid]) & fm[pix[valid]]            # :470-482
    idx = np.arange(xyz.shape[0], dtype=np.int64)
    return idx[~win], idx[win], pix[win], np.nonzero(~fm)[0].astype(np.int64)


def positional_encoding(positions: Tensor, freqs: int) -> Tensor:
    """encoder_freesplat.py:62-77 (ori=False): [..., D] -> [..., 2*D*freqs], per input value the
    frequencies 2^k ascending, (sin, cos) interleaved."""
    bands = (2 ** torch.arange(freqs).float()).to(positions.device)
    pts = (positions[..., None] * bands).reshape(positions.shape[:-1] + (freqs * positions.shape[-1],))
    return torch.stack([torch.sin(pts), torch.cos(pts)], dim=-1).reshape(pts.shape[:-1] + (pts.shape[-1] * 2,))


def gru(p: dict, x: Tensor, hid: Tensor, xe: Tensor, he: Tensor) -> Tensor:
    """networks.py:201-214.  p: {"mlp_z.0.weight", ...}; x, hid [n,64]; xe, he [n,24]."""
    def mlp(name, t):
        t = F.relu(F.linear(t, p[f"{name}.0.weight"], p[f"{name}.0.bias"]))
        return F.linear(t, p[f"{name}.2.weight"], p[f"{name}.2.bias"])
    x1 = torch.cat((x, xe), -1)
    h1 = torch.cat((hid, he), -1)
    cat = torch.cat((h1, x1), -1)
    r = torch.sigmoid(mlp("mlp_r", cat))
    z = torch.sigmoid(mlp("mlp_z", cat))
    q = torch.tanh(mlp("mlp_n", torch.cat((r * hid, x1), -1)))
    return (1 - z) * hid + z * q


def fuse_gaussians(gru_params: dict, latents: Tensor, coords: Tensor, densities: Tensor, weights: Tensor,
                   depths: Tensor, extrinsics: Tensor, intrinsics: Tensor, image_shape, depth_thres: float = 0.1,
                   w2c_all: Tensor | None = None):
    """latents [1,V,P,64], coords [1,V,P,1,1,3], densities/weights [1,V,P,1,1], depths [V,1,h,w],
    extrinsics [1,V,4,4] (or [V,4,4]), intrinsics [1,V,3,3] normalised.
    Returns (latent [1,M,64], xyz [1,M,3], extrinsics [1,M,4,4], depths [1,M]) -- encoder_freesplat.py:522."""
    h, w = image_shape
    V = latents.shape[1]
    P = h * w
    E = extrinsics.reshape(-1, 4, 4)
    Kn = intrinsics.reshape(-1, 3, 3)
    g = latents[0]                       # [V,P,64]
    xs = coords[0, :, :, 0, 0]           # [V,P,3]
    rho = densities[0, :, :, 0, 0]       # [V,P]
    om = weights[0, :, :, 0, 0]
    d = depths.reshape(V, P)
    G, X, R, O = g[0], xs[0], rho[0], om[0]
    Ex = E[0][None].repeat(P, 1, 1)
    Dp = d[0]
    for i in range(1, V):
        K = Kn[i].clone()
        K[:1] *= w                                                            # :446-447
        K[1:2] *= h
        kpix = np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]], f32)
        # (the reference: `extrinsic.inverse()`, encoder_freesplat.py:455.  `w2c_all` [V,4,4]: use these inverses instead --
        #  a pixel's round-half-even decision can hinge on the last bit of the matrix, and the LU inverse of a host LAPACK
        #  and of the GPU's solver differ there; full-size tests hand both sides the same matrices)
        w2c = (torch.linalg.inv(E[i]) if w2c_all is None else w2c_all[i]).numpy()
        keep, fuse, fpix, app = (torch.from_numpy(a) for a in
                                 match_step(X.detach().numpy(), w2c, kpix, d[i].detach().numpy(), h, w, depth_thres))
        if fuse.numel() > 0:                                                  # :484
            xe = positional_encoding(torch.stack([R[fuse], om[i][fpix]], -1), 6)       # :485
            he = positional_encoding(torch.stack([rho[i][fpix], O[fuse]], -1), 6)      # :486
            fused = gru(gru_params, g[i][fpix], G[fuse], xe, he)                        # :487-490
            w0, w1 = R[fuse], rho[i][fpix]
            Xf = (X[fuse] * w0[:, None] + xs[i][fpix] * w1[:, None]) / (w0 + w1)[:, None]       # :496-497
            Ef = (Ex[fuse] * w0[:, None, None] + E[i][None] * w1[:, None, None]) / (w0 + w1)[:, None, None]
            Df = (Dp[fuse] * w0 + d[i][fpix] * w1) / (w0 + w1)                                  # :505-506
            G = torch.cat([G[keep], fused]); X = torch.cat([X[keep], Xf])                       # :492
            R = torch.cat([R[keep], w0 + w1]); O = torch.cat([O[keep], O[fuse] + om[i][fpix]])  # :498-501
            Ex = torch.cat([Ex[keep], Ef]); Dp = torch.cat([Dp[keep], Df])
        G = torch.cat([G, g[i][app]]); X = torch.cat([X, xs[i][app]])                           # :508-519
        R = torch.cat([R, rho[i][app]]); O = torch.cat([O, om[i][app]])
        Ex = torch.cat([Ex, E[i][None].repeat(app.numel(), 1, 1)]); Dp = torch.cat([Dp, d[i][app]])
    return G[None], X[None], Ex[None], Dp[None]
