"""Pixel-wise Triplet Fusion: drop-in for EncoderFreeSplat.fuse_gaussians.

Mirrors /root/reference/src/model/encoder/encoder_freesplat.py:431-522 (same signature, same four
outputs in the same ORDER) together with the modules it needs: `GRU`
(src/model/encoder/modules/networks.py:188-214, parameter names mlp_{z,r,n}.{0,2}.{weight,bias} and
construction order kept so checkpoints and seeded inits carry over) and `positional_encoding`
(encoder_freesplat.py:62-77).

Split of work per fold step:
  * everything data-dependent and non-differentiable -- projection of the M global Gaussians, the
    per-pixel z-buffer, the depth-consistency mask, winner selection and the three ORDERED index
    lists -- is fs_ptf_match in libfreesplat_hip.so (6 launches, no host sync, replaces
    scatter_reduce_ + 2x torch.isin + boolean-mask indexing + 4 syncs of the reference);
  * the differentiable part (gather by those indices, GRU, density-weighted blends, concatenation)
    stays in torch ops on the device, so autograd gives the reference's gradients unchanged; the GRU's
    176/152 -> 64 -> 64 linears are plain library GEMMs (rocBLAS).
One host sync per view remains (the three list lengths size the torch tensors).
"""
from __future__ import annotations

import ctypes as C

import torch
from torch import Tensor, nn

from . import _lib


def positional_encoding(positions: Tensor, freqs: int, ori: bool = False) -> Tensor:
    bands = (2 ** torch.arange(freqs).float()).to(positions.device)
    ori_c = positions.shape[-1]
    pts = (positions[..., None] * bands).reshape(positions.shape[:-1] + (freqs * positions.shape[-1],))
    if ori:
        return torch.cat([positions, torch.sin(pts), torch.cos(pts)], dim=-1).reshape(
            pts.shape[:-1] + (pts.shape[-1] * 2 + ori_c,))
    return torch.stack([torch.sin(pts), torch.cos(pts)], dim=-1).reshape(pts.shape[:-1] + (pts.shape[-1] * 2,))


class GRU(nn.Module):
    def __init__(self, input_channel=64, hidden_channel=64, weights_dim=24):
        super().__init__()
        mk = lambda d: nn.Sequential(nn.Linear(d, hidden_channel), nn.ReLU(), nn.Linear(hidden_channel, hidden_channel))
        self.mlp_z = mk(hidden_channel + input_channel + 2 * weights_dim)
        self.mlp_r = mk(hidden_channel + input_channel + 2 * weights_dim)
        self.mlp_n = mk(hidden_channel + input_channel + 1 * weights_dim)

    def forward(self, input_feat, hidden_feat, input_weights_emb, hidden_weights_emb):
        if len(input_feat.size()) == 2 and input_feat.size(0) == 1:
            input_feat = input_feat.unsqueeze(1)
        if hidden_feat is None:
            hidden_feat = torch.zeros_like(input_feat)
        x1 = torch.cat((input_feat, input_weights_emb), dim=-1)
        h1 = torch.cat((hidden_feat, hidden_weights_emb), dim=-1)
        cat = torch.cat((h1, x1), dim=-1)
        r = torch.sigmoid(self.mlp_r(cat))
        z = torch.sigmoid(self.mlp_z(cat))
        q = torch.tanh(self.mlp_n(torch.cat((r * hidden_feat, x1), dim=-1)))
        return (1 - z) * hidden_feat + z * q


def match_view(xyz: Tensor, w2c: Tensor, kpix: Tensor, depth_i: Tensor, h: int, w: int, depth_thres: float = 0.1):
    """fs_ptf_match on device tensors: xyz [M,3], w2c [4,4], kpix [4], depth_i [h*w] -> ascending int64
    index tensors (keep_idx, fuse_idx, fuse_pix, append_pix).  One host sync (the three counts)."""
    if xyz.device.type != "cuda":
        raise RuntimeError(f"freesplat_amd PTF: tensors must live on a HIP device (got {xyz.device}); no CPU path")
    dev = xyz.device
    M, P = xyz.shape[0], h * w
    L = _lib.lib()
    xyz = xyz.detach().float().contiguous()
    scratch = torch.empty(L.fs_ptf_scratch_bytes(M, h, w), dtype=torch.uint8, device=dev)
    keep = torch.empty(max(M, 1), dtype=torch.int64, device=dev)
    fuse = torch.empty(max(M, 1), dtype=torch.int64, device=dev)
    fpix = torch.empty(max(M, 1), dtype=torch.int64, device=dev)
    app = torch.empty(P, dtype=torch.int64, device=dev)
    counts = torch.empty(4, dtype=torch.int32, device=dev)
    p = _lib.ptr
    w2c_, kpix_, depth_ = (t.detach().float().contiguous() for t in (w2c, kpix, depth_i))  # alive until the launch is queued
    _lib.check(L.fs_ptf_match(M, h, w, p(xyz), p(w2c_), p(kpix_), p(depth_),
                              C.c_float(depth_thres), p(scratch), p(keep), p(fuse), p(fpix), p(app), p(counts),
                              _lib.current_stream()), "fs_ptf_match")
    nk, nf, na, _ = counts.tolist()
    return keep[:nk], fuse[:nf], fpix[:nf], app[:na]


_table_cache: dict = {}
LAST_FOLD_COUNTS = None   # device tensor [V,4] (kept, fused, appended, state rows) of the last fused fold: bench accounting


def gru_tables(gru: "GRU") -> Tensor:
    """The GRU's weights and biases in the MFMA operand order of csrc/ptf_gru.hip: rows of 64 lanes,
    lane l = (p = l & 31, hf = l >> 5).  Cached per parameter version."""
    params = [gru.mlp_r[0].weight, gru.mlp_r[0].bias, gru.mlp_r[2].weight, gru.mlp_r[2].bias,
              gru.mlp_z[0].weight, gru.mlp_z[0].bias, gru.mlp_z[2].weight, gru.mlp_z[2].bias,
              gru.mlp_n[0].weight, gru.mlp_n[0].bias, gru.mlp_n[2].weight, gru.mlp_n[2].bias]
    key = (id(gru), tuple((q.data_ptr(), q._version) for q in params))
    hit = _table_cache.get(id(gru))
    if hit is not None and hit[0] == key:
        return hit[1]
    dev = params[0].device
    with torch.no_grad():
        Wr1, br1, Wr2, br2, Wz1, bz1, Wz2, bz2, Wn1, bn1, Wn2, bn2 = [q.detach().float() for q in params]
        lane = torch.arange(64, device=dev)
        pp, hf = lane & 31, lane >> 5
        acc_row = lambda q, h: (q & 3) + 8 * (q >> 2) + 4 * h
        unit = lambda s: acc_row(s[:, None] & 15, hf[None, :]) + 32 * (s[:, None] >> 4)      # [steps, 64]

        def l1(W):   # [64,176]: step s <-> input s + 88*hf
            s = torch.arange(88, device=dev)
            col = s[:, None] + 88 * hf[None, :]
            return torch.stack([W[32 * b + pp[None, :].expand_as(col), col] for b in range(2)])

        def l2(W):   # [64,64]: step s <-> hidden unit of accumulator register s & 15, block s >> 4
            col = unit(torch.arange(32, device=dev))
            return torch.stack([W[32 * b + pp[None, :].expand_as(col), col] for b in range(2)])

        def n1(W):   # [64,152]: steps 0..31 <-> r*hid units, 32..75 <-> x|xe input 64 + (s-32) + 44*hf
            s = torch.arange(44, device=dev)
            col = torch.cat([unit(torch.arange(32, device=dev)), 64 + s[:, None] + 44 * hf[None, :]])
            return torch.stack([W[32 * b + pp[None, :].expand_as(col), col] for b in range(2)])

        def bias(bv):  # [64] -> [2,16,64]
            q = torch.arange(16, device=dev)
            return torch.stack([bv[acc_row(q[:, None], hf[None, :]) + 32 * b] for b in range(2)])

        tab = torch.cat([t.reshape(-1, 64) for t in (l1(Wr1), l1(Wz1), l2(Wr2), l2(Wz2), n1(Wn1), l2(Wn2),
                                                     bias(br1), bias(bz1), bias(br2), bias(bz2), bias(bn1),
                                                     bias(bn2))]).contiguous()
    assert tab.shape[0] == _lib.lib().fs_ptf_gru_table_rows()
    _table_cache[id(gru)] = (key, tab)
    return tab


def _fuse_gaussians_fused(gru, gaussians, coords, densities, weight_emb, depths, extrinsics, intrinsics, image_shape,
                          depth_thres):
    """Inference path (no autograd): ONE library call (fs_ptf_fold) folds all views -- per view match -> GRU inputs ->
    GRU on the fp32 matrix cores -> next state, with every data-dependent size kept on the device -- and the host
    syncs once afterwards, for the final number of Gaussians (the reference syncs four times per view).
    Same results and order as the differentiable path below."""
    L = _lib.lib()
    p = _lib.ptr
    h, w = image_shape
    V = gaussians[0].shape[1]
    f = lambda t: t.detach().float().contiguous()
    lat = f(gaussians[0][0])                      # [V,P,64]
    xs = f(coords[0][0, :, :, 0, 0])              # [V,P,3]
    rho = f(densities[0, :, :, 0, 0])             # [V,P]
    om = f(weight_emb[0, :, :, 0, 0])
    dep = f(depths.reshape(V, -1))
    Es = f(extrinsics[0]).reshape(V, 16)          # [V,16]
    dev = lat.device
    P = h * w
    if V == 1:
        return lat[:1], xs[:1], Es[0].reshape(1, 1, 4, 4).repeat(1, P, 1, 1), dep[:1]
    tables = gru_tables(gru)
    # the state after view 0 is view 0 itself; fs_ptf_fold folds views 1 .. V-1 into it, writing the successive states
    # alternately into two sets of buffers -- one library call (camera constants included), one host sync afterwards
    Kn = f(intrinsics[0]).reshape(V, 9)
    w2c = torch.linalg.inv_ex(Es.view(V, 4, 4)).inverse.reshape(V, 16).contiguous()   # (the reference's own inverse)
    rows = 2 * P if V == 2 else V * P
    bufs = [[torch.empty(rows, n, device=dev) for n in (64, 3, 1, 1, 16, 1)] for _ in range(1 if V == 2 else 2)]
    ptrs = [(C.c_void_p * 6)(*[t.data_ptr() for t in b]) for b in bufs]
    counts = torch.empty(V, 4, dtype=torch.int32, device=dev)
    scratch = torch.empty(L.fs_ptf_fold_bytes(V, h, w), dtype=torch.uint8, device=dev)
    _lib.check(L.fs_ptf_fold(V, h, w, p(lat), p(xs), p(rho), p(om), p(dep), p(Es), p(w2c), p(Kn), C.c_float(depth_thres),
                             p(tables), p(scratch), ptrs[0], ptrs[-1], p(counts), _lib.current_stream()), "fs_ptf_fold")
    global LAST_FOLD_COUNTS
    LAST_FOLD_COUNTS = counts
    n = int(counts[V - 1, 3].item())               # the only host sync of the fold
    G, X, _, _, E, D = bufs[0] if ((V - 1) & 1 or V == 2) else bufs[1]
    return G[None, :n], X[None, :n], E[:n].view(1, n, 4, 4), D[None, :n, 0]


def fuse_gaussians(self, gaussians, coords, densities, weight_emb, depths, extrinsics, intrinsics, image_shape,
                   depth_thres=0.1):
    """Same contract as EncoderFreeSplat.fuse_gaussians (encoder_freesplat.py:431-522); `self` only
    needs a `.gru` attribute.  gaussians = [latents [1,V,P,64]], coords = [[1,V,P,1,1,3]],
    densities / weight_emb [1,V,P,1,1], depths [V,1,h,w], extrinsics [1,V,4,4], intrinsics [1,V,3,3].
    Returns (latents [1,M,64], xyz [1,M,3], extrinsics [1,M,4,4], depths [1,M]).
    Without autograd (eval / torch.no_grad) the fold runs through the fused HIP data-movement kernels."""
    needs_grad = torch.is_grad_enabled() and (
        any(t.requires_grad for t in (gaussians[0], coords[0], densities, weight_emb, depths))
        or any(q.requires_grad for q in self.gru.parameters()))
    if not needs_grad and gaussians[0].shape[0] == 1:
        return _fuse_gaussians_fused(self.gru, gaussians, coords, densities, weight_emb, depths, extrinsics, intrinsics,
                                     image_shape, depth_thres)
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
        w2c = torch.linalg.inv_ex(extrinsic).inverse
        keep, fuse, fpix, app = match_view(X[0], w2c, kpix, depths[i], h, w, depth_thres)
        if fuse.numel() > 0:
            xe = positional_encoding(torch.cat([R[:, fuse], weight_emb[:, i, fpix]], dim=-1), 6)
            he = positional_encoding(torch.cat([densities[:, i, fpix], O[:, fuse]], dim=-1), 6)
            fused = self.gru(gaussians[0][:, i, fpix].unsqueeze(2), G[:, fuse].unsqueeze(2), xe, he).squeeze(2)
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


class PixelwiseTripletFusion(nn.Module):
    """Owner of the `gru` sub-module with the reference's `fuse_gaussians` method bound to it, so that
    `encoder.gru.*` state-dict keys and `encoder.fuse_gaussians(...)` call sites carry over."""

    def __init__(self):
        super().__init__()
        self.gru = GRU()

    fuse_gaussians = fuse_gaussians
