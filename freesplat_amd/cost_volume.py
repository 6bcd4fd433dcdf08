"""Drop-in for the reference's plane-sweep cost volume module.

Mirrors /root/reference/src/model/encoder/modules/cost_volume.py:
    AVGFeatureVolumeManager(matching_height, matching_width, num_depth_bins=64,
                            mlp_channels=[202,32,32,1], matching_dim_size=16)       (:399-426)
    .forward(cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK, min_depth,
             max_depth, depth_planes_bdhw=None, return_mask=False) -> [B, D, h, w]     (:351-381)
    .generate_depth_planes(batch_size, min_depth, max_depth)                           (:98-134)
with the same parameter / buffer names (so the reference's checkpoints load: `linear_ramp_1d11`,
`backprojector.pix_coords_13N`, `projector.eps`, `mlp.net.{0,2,4}.{weight,bias}`) and the same
construction order (so a seeded construction yields the same initial weights).  The compute is
fs_cost_volume_forward in libfreesplat_hip.so -- no torch fallback.
"""
from __future__ import annotations

import os

import torch
from torch import Tensor, nn

from . import _lib


def save_activations(K: int) -> bool:
    """Whether a training forward keeps the MLP's input of every (view, plane, pixel) point for the backward
    (fs_cost_volume_forward_train: C + 2 floats per point, what autograd keeps of the averaged features) or the backward
    gathers the K sources' taps again.  Default: from K = 4 sources per view up; FREESPLAT_CV_SAVE=0 / 1 forces either (read at
    every call).  Measured in round 6 (profiles/r6_cv_saved_ab.txt; rocprofv3, forward sweep + pass 1 of the backward): 10 views,
    K = 8: 3.56 + 6.06 ms recomputed, 4.74 + 2.75 ms saved (training step 15.3 -> 13.4 ms); config-3 scale, K = 2: 2.75 + 6.80 against
    3.89 + 6.30 (13.0 -> 13.5 ms); native, K = 1: 1.32 -> 1.43 ms -- the write costs the forward ~1.1 ms per 3 - 6 GB, the gather
    costs pass 1 ~0.25 ms per source at config-3 scale and ~0.4 ms per source in the tap-bound K = 8 regime.  (Rounds 4 - 5 read
    torch.is_grad_enabled() inside the Function's forward -- always False there -- so the saved path never ran through this module
    and its "buys nothing" was an A/A measurement; the module's forward now decides, in the caller's grad mode.)"""
    e = os.environ.get("FREESPLAT_CV_SAVE")
    if e in ("0", "1"):
        return e == "1"
    return K >= 4


class _Backprojector(nn.Module):
    """Holds the `pix_coords_13N` buffer of sr_utils/geometry_utils.py:22-48 (state-dict parity);
    the kernel regenerates (u+0.5, v+0.5, 1) itself."""

    def __init__(self, height: int, width: int):
        super().__init__()
        xx, yy = torch.meshgrid(torch.arange(width), torch.arange(height), indexing="xy")
        pix = torch.stack((xx, yy), 0) + 0.5
        pix = torch.cat([pix, torch.ones_like(pix[:1])], 0).flatten(1).unsqueeze(0)
        self.register_buffer("pix_coords_13N", pix)


class _Projector(nn.Module):
    def __init__(self, eps: float = 1e-8):
        super().__init__()
        self.register_buffer("eps", torch.tensor(eps).view(1, 1, 1))


class MLP(nn.Module):
    """networks.py:218-236: Linear / LeakyReLU stack, final activation disabled."""

    def __init__(self, channel_list, disable_final_activation: bool = False):
        super().__init__()
        layers = []
        for i in range(len(channel_list) - 1):
            layers.append(nn.Linear(channel_list[i], channel_list[i + 1]))
            layers.append(nn.LeakyReLU(inplace=True))
        if disable_final_activation:
            layers = layers[:-1]
        self.net = nn.Sequential(*layers)


CALLS = {"forward_train": 0}     # how often the activation-keeping forward ran (tests: the path is alive, and only when a backward can follow)


class _CostVolumeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cur_feats, src_feats, src_extrinsics, src_Ks, cur_invK, planes, strides, w1, b1, w2, b2,
                w3, b3, layout=0, train=False):
        B, K, C, h, w = src_feats.shape
        D = planes.shape[0] if planes.dim() == 1 else planes.shape[1]
        dev = cur_feats.device
        L = _lib.lib()
        ws = torch.empty(L.fs_cost_volume_workspace_bytes(B, K, C, h, w), dtype=torch.uint8, device=dev)
        out = torch.empty(B, D, h, w, dtype=torch.float32, device=dev)
        p = _lib.ptr
        # A backward will follow: with many sources per view the training forward keeps the MLP's input of every point and
        # the backward starts from it instead of gathering all taps again (save_activations)
        # (`train` comes from the caller: INSIDE a Function's forward torch.is_grad_enabled() is always False, and
        #  ctx.needs_input_grad reports the parameters' requires_grad even under no_grad)
        train = bool(train) and strides[2] == 0 and K <= 16 and save_activations(K)
        if train:
            CALLS["forward_train"] += 1
            saved = torch.empty(L.fs_cost_volume_saved_bytes(B, C, h, w, D), dtype=torch.uint8, device=dev)
            _lib.check(L.fs_cost_volume_forward_train(B, K, C, h, w, D, p(cur_feats), p(src_feats), p(src_extrinsics),
                                                      p(src_Ks), p(cur_invK), p(planes), strides[0], strides[1], strides[2],
                                                      p(w1), p(b1), p(w2), p(b2), p(w3), p(b3), p(ws), p(out), p(saved),
                                                      _lib.current_stream()), "fs_cost_volume_forward_train")
        elif layout:
            # inference on channels_last maps (pixel-major records): read in place, no re-layout pass (AVGFeatureVolumeManager.forward)
            saved = None
            _lib.check(L.fs_cost_volume_forward_layout(B, K, C, h, w, D, p(cur_feats), p(src_feats), p(src_extrinsics),
                                                       p(src_Ks), p(cur_invK), p(planes), strides[0], strides[1], strides[2],
                                                       p(w1), p(b1), p(w2), p(b2), p(w3), p(b3), p(ws), p(out), layout,
                                                       _lib.current_stream()), "fs_cost_volume_forward_layout")
            return out
        else:
            saved = None
            _lib.check(L.fs_cost_volume_forward(B, K, C, h, w, D, p(cur_feats), p(src_feats), p(src_extrinsics),
                                                p(src_Ks), p(cur_invK), p(planes), strides[0], strides[1], strides[2],
                                                p(w1), p(b1), p(w2), p(b2), p(w3), p(b3), p(ws), p(out),
                                                _lib.current_stream()), "fs_cost_volume_forward")
        ctx.save_for_backward(cur_feats, src_feats, src_extrinsics, src_Ks, cur_invK, planes, w1, b1, w2, b2, w3)
        ctx.strides, ctx.D, ctx.saved = strides, D, saved
        return out

    @staticmethod
    def backward(ctx, g):
        cur_feats, src_feats, src_extrinsics, src_Ks, cur_invK, planes, w1, b1, w2, b2, w3 = ctx.saved_tensors
        B, K, C, h, w = src_feats.shape
        D, strides = ctx.D, ctx.strides
        dev = g.device
        L = _lib.lib()
        ws = torch.empty(L.fs_cost_volume_backward_workspace_bytes_for(B, K, C, h, w, D, strides[2]), dtype=torch.uint8, device=dev)
        d_cur, d_src = torch.empty_like(cur_feats), torch.empty_like(src_feats)
        e = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        d_w1, d_b1, d_w2, d_b2, d_w3, d_b3 = e(32, C + 1), e(32), e(32, 32), e(32), e(1, 32), e(1)
        p = _lib.ptr
        g_ = g.contiguous()
        if ctx.saved is not None:
            _lib.check(L.fs_cost_volume_backward_train(B, K, C, h, w, D, p(cur_feats), p(src_feats), p(src_extrinsics),
                                                       p(src_Ks), p(cur_invK), p(planes), strides[0], strides[1], strides[2],
                                                       p(w1.detach()), p(b1.detach()), p(w2.detach()), p(b2.detach()),
                                                       p(w3.detach()), p(g_), p(ws), p(ctx.saved), p(d_cur), p(d_src),
                                                       p(d_w1), p(d_b1), p(d_w2), p(d_b2), p(d_w3), p(d_b3),
                                                       _lib.current_stream()), "fs_cost_volume_backward_train")
            ctx.saved = None
        else:
            _lib.check(L.fs_cost_volume_backward(B, K, C, h, w, D, p(cur_feats), p(src_feats), p(src_extrinsics),
                                                 p(src_Ks), p(cur_invK), p(planes), strides[0], strides[1], strides[2],
                                                 p(w1.detach()), p(b1.detach()), p(w2.detach()), p(b2.detach()),
                                                 p(w3.detach()), p(g_), p(ws), p(d_cur), p(d_src), p(d_w1), p(d_b1),
                                                 p(d_w2), p(d_b2), p(d_w3), p(d_b3), _lib.current_stream()),
                       "fs_cost_volume_backward")
        # (every gradient, the MLP's included, comes out of the one kernel: no per-point workspace, no GEMMs here)
        return d_cur, d_src, None, None, None, None, None, d_w1, d_b1, d_w2, d_b2, d_w3, d_b3, None, None


def _dev32(t: Tensor, name: str) -> Tensor:
    if t.device.type != "cuda":
        raise RuntimeError(f"freesplat_amd cost volume: `{name}` must live on a HIP device (got {t.device}); "
                           "there is no CPU path")
    return t.float().contiguous()


def _pixel_major(t: Tensor) -> bool:
    """A float32 HIP tensor [..., C, h, w] whose MEMORY is [..., h, w, C] (channels_last for 4-D; the same strides under leading
    dimensions for the 5-D source maps) and not also plain-contiguous (C = 1 or h * w = 1 are both at once: nothing to gain)."""
    if t.device.type != "cuda" or t.dtype != torch.float32 or t.dim() < 4 or t.is_contiguous():
        return False
    return t.movedim(-3, -1).is_contiguous()


class AVGFeatureVolumeManager(nn.Module):
    def __init__(self, matching_height, matching_width, num_depth_bins=64, mlp_channels=[202, 32, 32, 1],
                 matching_dim_size=16):
        super().__init__()
        self.num_depth_bins = num_depth_bins
        self.matching_height = matching_height
        self.matching_width = matching_width
        self.register_buffer("linear_ramp_1d11", torch.linspace(0, 1, num_depth_bins).view(1, num_depth_bins, 1, 1))
        self.backprojector = _Backprojector(matching_height, matching_width)
        self.projector = _Projector()
        mlp_channels = list(mlp_channels)  # (the reference mutates its default argument, :423)
        mlp_channels[0] = matching_dim_size + 1
        if mlp_channels[1:] != [32, 32, 1]:
            raise NotImplementedError("the HIP cost volume implements the shipped 32-32-1 MLP")
        self.mlp = MLP(channel_list=mlp_channels, disable_final_activation=True)

    def generate_depth_planes(self, batch_size: int, min_depth: Tensor, max_depth: Tensor) -> Tensor:
        ramp = self.linear_ramp_1d11.expand(batch_size, self.num_depth_bins, 1, 1)
        inv_min, inv_max = 1 / min_depth, 1 / max_depth
        planes = 1 / (inv_min + (inv_max - inv_min) * ramp)
        planes = planes.expand(batch_size, self.num_depth_bins, self.matching_height, self.matching_width)
        self.depth_planes_bdhw = planes
        return planes

    def forward(self, cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK, min_depth, max_depth,
                depth_planes_bdhw=None, return_mask=False):
        B, K, C, h, w = src_feats.shape
        if (h, w) != (self.matching_height, self.matching_width):
            raise RuntimeError("feature maps do not match matching_height/width")
        if depth_planes_bdhw is None and min_depth.numel() == 1 and max_depth.numel() == 1:
            # the call of encoder_freesplat.py:280-288 (near / far of the first view): generate_depth_planes in ONE launch,
            # rounded operation by operation as the module's eight elementwise torch launches round it
            D = self.num_depth_bins
            flat = torch.empty(D, dtype=torch.float32, device=src_feats.device)
            md, Md = _dev32(min_depth.reshape(1), "min_depth"), _dev32(max_depth.reshape(1), "max_depth")
            ramp = _dev32(self.linear_ramp_1d11.reshape(D), "linear_ramp_1d11")
            p = _lib.ptr
            _lib.check(_lib.lib().fs_cost_volume_depth_planes(D, p(md), p(Md), p(ramp), p(flat), _lib.current_stream()),
                       "fs_cost_volume_depth_planes")
            self.depth_planes_bdhw = flat.view(1, D, 1, 1).expand(B, D, h, w)     # (the attribute the reference leaves, :133)
            strides = (0, 1, 0)
        elif depth_planes_bdhw is None:
            planes = self.generate_depth_planes(B, min_depth, max_depth)
            flat = _dev32(planes[0, :, 0, 0], "depth planes")   # [D]: identical for every batch row / pixel
            if planes.shape[0] > 1 and planes.stride(0) != 0:
                flat = _dev32(planes[:, :, 0, 0], "depth planes")
                strides = (flat.shape[1], 1, 0)
            else:
                strides = (0, 1, 0)
        else:
            flat = _dev32(depth_planes_bdhw, "depth_planes_bdhw")
            D = flat.shape[1]
            strides = (D * h * w, h * w, 1)
        net = self.mlp.net
        # channels_last feature maps ARE the pixel-major records the K >= 2 sweep gathers from: an inference call reads them in
        # place (fs_cost_volume_forward_layout) instead of paying a re-layout pass per map -- 425 of the 755 MB a 10-view K = 8 call
        # moves.  With autograd on, the maps go through .contiguous() as before (the backward takes [C, h, w] maps).
        # a backward can follow this call: grad mode on (read HERE, outside the autograd Function) and something requires grad
        train = torch.is_grad_enabled() and (cur_feats.requires_grad or src_feats.requires_grad
                                             or any(q.requires_grad for q in self.mlp.parameters()))
        layout = 0
        if not train:
            layout = ((1 if _pixel_major(cur_feats) else 0) | (2 if _pixel_major(src_feats) else 0))
        if layout:
            cf = cur_feats if layout & 1 else _dev32(cur_feats, "cur_feats")
            sf = src_feats if layout & 2 else _dev32(src_feats, "src_feats")
            return _CostVolumeFn.apply(cf, sf, _dev32(src_extrinsics, "src_extrinsics"), _dev32(src_Ks, "src_Ks"),
                                       _dev32(cur_invK, "cur_invK"), flat, strides, net[0].weight, net[0].bias,
                                       net[2].weight, net[2].bias, net[4].weight, net[4].bias, layout)
        return _CostVolumeFn.apply(_dev32(cur_feats, "cur_feats"), _dev32(src_feats, "src_feats"),
                                   _dev32(src_extrinsics, "src_extrinsics"), _dev32(src_Ks, "src_Ks"),
                                   _dev32(cur_invK, "cur_invK"), flat, strides, net[0].weight, net[0].bias,
                                   net[2].weight, net[2].bias, net[4].weight, net[4].bias, 0, train)


def sharded_cost_volume(manager, local_feats: Tensor, extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor,
                        image_hw: tuple[int, int], num_context_views: int, group=None) -> Tensor:
    """Plane-sweep cost volumes of the context views of ONE scene with the views sharded over a process group
    (the `B = b*V` rows of encoder_freesplat.py:260-288 are independent, b = 1 in every shipped
    config).  Rank r holds `local_feats` [V_r, C, h/4, w/4] -- the matching features of ITS views
    (view_sharding.shard_range(V, r, world)) -- and the full cameras extrinsics [1,V,4,4] / intrinsics [1,V,3,3] /
    near, far [1,V].  One all-gather of the feature maps (2.36 MB per view at the native 96x128; its backward is a
    reduce-scatter), then every rank sweeps only its own current views: returns [V_r, D, h/4, w/4], which stays local
    for the per-view CNN that consumes it.  `manager` = AVGFeatureVolumeManager (any callable with its forward's
    keyword arguments).  Without an initialised process group (or world size 1) this is the unsharded call."""
    import torch.distributed as dist
    from .encoder_glue import prepare_cost_volume_inputs
    from .view_sharding import gather_features_autograd, shard_range
    if extrinsics.shape[0] != 1:
        raise NotImplementedError("sharded_cost_volume: one scene per call (b = 1, config/main.yaml:26)")
    V = extrinsics.shape[1]
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        feats, mine = local_feats, range(V)
    else:
        mine = shard_range(V, dist.get_rank(group), world)
        if local_feats.shape[0] != len(mine):
            raise ValueError(f"rank holds {local_feats.shape[0]} feature maps, expected {len(mine)}")
        feats = gather_features_autograd(local_feats, V, group)
    if len(mine) == 0:
        D = getattr(manager, "num_depth_bins", 0)
        return feats.new_zeros((0, D) + tuple(feats.shape[-2:])) + 0.0 * feats.sum()
    # only this rank's rows of the B = V dimension are materialised (sources indexed out of the gathered maps)
    return manager(**prepare_cost_volume_inputs(extrinsics, intrinsics, feats, near, far, image_hw, num_context_views,
                                                rows=mine))
