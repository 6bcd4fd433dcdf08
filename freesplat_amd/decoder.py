"""Decoder boundary: camera framing + per-view rasterizer calls.

Host-side mirror of the reference's
  /root/reference/src/model/decoder/cuda_splatting.py:17-132  (get_projection_matrix + render_cuda; the camera framing of
                                                               :17-44 / :64-87 and get_fov, projection.py:233-247, are ONE
                                                               HIP launch here: frame_views -> fs_frame_views)
  /root/reference/src/model/decoder/decoder_splatting_cuda.py:35-75 (DecoderSplattingCUDA.forward)
with the same names, argument meaning and results, so that a caller of `render_cuda` /
`DecoderSplattingCUDA` can switch over unchanged.  Everything here is plumbing in torch; the
compute is libfreesplat_hip.so via freesplat_amd.rasterizer.

Beyond the reference: `render_views` renders v target views of ONE shared Gaussian set without
materialising v copies of it (decoder_splatting_cuda.py:55-58 `repeat`) and with a single host
sync per call instead of three per view; gradients of the shared set are accumulated on device.
"""
from __future__ import annotations

from dataclasses import dataclass
from math import isqrt
from typing import Optional

import ctypes as C
import os

import torch
from torch import Tensor, nn

from . import rasterizer as R
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer


_const_cache: dict = {}


def _consts(device: torch.device) -> dict:
    """Small constant tensors kept resident per device: creating them per call costs a pageable
    host-to-device copy each, and on ROCm that copy waits for all queued work of the stream."""
    key = (device.type, device.index)
    c = _const_cache.get(key)
    if c is None:
        row, col = torch.triu_indices(3, 3)
        c = _const_cache[key] = dict(
            triu_row=row.to(device), triu_col=col.to(device))
    return c


def frame_views(extrinsics, intrinsics, near, far, scale_invariant: bool = True):
    """The per-view camera matrices of cuda_splatting.py:64-87 (get_fov, projection.py:233-247; get_projection_matrix,
    cuda_splatting.py:17-44; the 1/near rescale; view = inverse(extrinsics)^T; full = view @ projection^T) for v views in one
    HIP launch (fs_frame_views, csrc/framing.hip: formed in double, each entry rounded once): returns (campos [v,3], scale [v],
    tanfov [v,2], view [v,4,4], full [v,4,4]), all on the device -- nothing of it is needed on the host by render_views.
    (A torch restatement of the reference's fp32 chain lives in tests/util_framing.py as the checker.)"""
    v = extrinsics.shape[0]
    dev = extrinsics.device
    if dev.type != "cuda":
        raise RuntimeError(f"freesplat_amd frame_views: tensors must live on a HIP device (got {dev}); no CPU path")
    f = lambda t: t.detach().to(torch.float32).contiguous()
    out = [torch.empty(v, n, dtype=torch.float32, device=dev) for n in (16, 16, 3, 2)] + [torch.empty(v, dtype=torch.float32, device=dev)]
    view, full, campos, tanfov, scale = out
    p = R._lib.ptr
    # (converted copies must stay referenced until the launch is queued: a temporary freed between two conversions
    # would hand its block to the next one)
    e_, k_, n_, f_ = f(extrinsics), f(intrinsics), f(near), f(far)
    R._lib.check(R._lib.lib().fs_frame_views(v, p(e_), p(k_), p(n_), p(f_), 1 if scale_invariant else 0, p(view), p(full),
                                             p(campos), p(tanfov), p(scale), R._lib.current_stream()), "fs_frame_views")
    return campos, scale, tanfov, view.view(v, 4, 4), full.view(v, 4, 4)


def render_cuda(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor,
                image_shape: tuple[int, int], background_color: Tensor, gaussian_means: Tensor,
                gaussian_covariances: Tensor, gaussian_sh_coefficients: Tensor,
                gaussian_opacities: Tensor, scale_invariant: bool = True, use_sh: bool = True):
    """Drop-in for cuda_splatting.py:47-132 (name kept for the drop-in).  One rasterizer call per
    batch element, each with its own copy of the Gaussians, like the reference.  Returns
    (color [B,3,H,W], depth [B,1,H,W])."""
    assert use_sh or gaussian_sh_coefficients.shape[-1] == 1
    campos, scale, tanfov, view, full = frame_views(extrinsics, intrinsics, near, far, scale_invariant)
    if scale_invariant:
        gaussian_covariances = gaussian_covariances * (scale[:, None, None, None] ** 2)
        gaussian_means = gaussian_means * scale[:, None, None]
    n = gaussian_sh_coefficients.shape[-1]
    degree = isqrt(n) - 1
    shs = gaussian_sh_coefficients.transpose(-1, -2).contiguous()  # b g xyz n -> b g n xyz
    b = view.shape[0]
    h, w = image_shape
    tan_h = tanfov.tolist()  # one sync for all views (reference: 2 per view)
    row, col = _consts(view.device)["triu_row"], _consts(view.device)["triu_col"]
    images, depths = [], []
    for i in range(b):
        mean_gradients = torch.zeros_like(gaussian_means[i], requires_grad=True)
        settings = GaussianRasterizationSettings(
            image_height=h, image_width=w, tanfovx=tan_h[i][0], tanfovy=tan_h[i][1],
            bg=background_color[i], scale_modifier=1.0, viewmatrix=view[i], projmatrix=full[i],
            sh_degree=degree, campos=campos[i], prefiltered=False, debug=False)
        rasterizer = GaussianRasterizer(settings)
        image, radii, depth, _ = rasterizer(
            means3D=gaussian_means[i], means2D=mean_gradients,
            shs=shs[i] if use_sh else None,
            colors_precomp=None if use_sh else shs[i, :, 0, :],
            opacities=gaussian_opacities[i, ..., None],
            cov3D_precomp=gaussian_covariances[i, :, row, col])
        images.append(image)
        depths.append(depth.unsqueeze(0))
    return torch.stack(images), torch.stack(depths)


def depth_to_relative_disparity(depth: Tensor, near: Tensor, far: Tensor, eps: float = 1e-10) -> Tensor:
    """src/model/encoder/epipolar/conversions.py:17-27: 0 at near, 1 at far."""
    disp_near, disp_far, disp = 1 / (near + eps), 1 / (far + eps), 1 / (depth + eps)
    return 1 - (disp - disp_far) / (disp_near - disp_far + eps)


def render_depth_cuda(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor, image_shape: tuple[int, int],
                      gaussian_means: Tensor, gaussian_covariances: Tensor, gaussian_opacities: Tensor,
                      scale_invariant: bool = True, mode: str = "depth") -> Tensor:
    """Drop-in for cuda_splatting.py:238-280 (dead code in the reference's own forward, decoder_splatting_cuda.py:66-73, kept
    for callers of it): the camera-space depth of every Gaussian -- or its disparity / relative disparity / log,
    `mode` -- rendered as a pre-computed colour (`colors_precomp` path of the rasterizer, black background) and averaged
    over the three channels.  [B,4,4] ... -> [B,H,W]."""
    homog = torch.cat([gaussian_means, torch.ones_like(gaussian_means[..., :1])], dim=-1)
    fake = torch.einsum("bij,bgj->bgi", torch.linalg.inv_ex(extrinsics).inverse, homog)[..., 2]
    if mode == "disparity":
        fake = 1 / fake
    elif mode == "relative_disparity":
        fake = depth_to_relative_disparity(fake, near[:, None], far[:, None])
    elif mode == "log":
        fake = fake.minimum(near[:, None]).maximum(far[:, None]).log()      # (the reference's own clamp order, :262)
    elif mode != "depth":
        raise ValueError(f"unknown depth rendering mode {mode!r}")
    b = fake.shape[0]
    result, _ = render_cuda(extrinsics, intrinsics, near, far, image_shape,
                            torch.zeros((b, 3), dtype=fake.dtype, device=fake.device), gaussian_means, gaussian_covariances,
                            fake[..., None, None].expand(*fake.shape, 3, 1), gaussian_opacities,
                            scale_invariant=scale_invariant, use_sh=False)
    return result.mean(dim=1)


# ---------------------------------------------------------------------------------------------
# Batched multi-view path: one shared Gaussian set, v views, one sync.
# ---------------------------------------------------------------------------------------------
# Set by view_sharding.GradExchange("chunked").install(): _RenderViews.backward then hands the Gaussian gradients over chunk by
# chunk of the rows while it is still producing them (object with begin / chunk_rows / chunk_ready).
GRAD_EXCHANGE_HOOK = None
_pending_checks: list = []  # (counters [v,2], states | None, device, (H, W)) of render_views(..., check="deferred") calls


def check_deferred() -> None:
    """Validate the instance-capacity counters of every deferred render_views call (one host sync).
    Raises if any of them overflowed -- their images are invalid and must be re-rendered."""
    global _pending_checks
    pend, _pending_checks = _pending_checks, []
    if not pend:
        return
    counters = torch.cat([c for c, _, _, _ in pend]).tolist()
    bad, k = 0, 0
    for c, states, dev, hw in pend:
        st = R._state(dev)
        for i in range(c.shape[0]):
            n_inst, overflow = counters[k]
            k += 1
            n_inst &= 0xFFFFFFFF
            if states is not None:
                states[i].num_rendered = n_inst
            st.last_instances = max(st.last_instances, n_inst)
            if overflow:    # (non-zero = the largest tile list: sizes the next call's capacity)
                bad += 1
                st.note_overflow(n_inst, overflow & 0xFFFFFFFF, *hw)
        if not bad:
            st.note_fit(*hw)
    if bad:
        raise R._lib.FreeSplatHipError(f"{bad} deferred view(s) overflowed their instance capacity; "
                                       "re-render them (capacity history has been updated)")


class _RenderViews(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, cov6, shs, opac, views, fulls, campos, bgs, tanfov, scale, h, w, degree,
                deferred):
        # cov6 = covariances [G,3,3], shs = harmonics [G,3,d_sh]: the reference's own layouts, read in place
        v = views.shape[0]
        N = means.shape[0]
        dev = means.device
        st = R._state(dev)
        cap = R.default_capacity(N, st, h, w)
        color = torch.empty(v, 3, h, w, dtype=torch.float32, device=dev)
        depth = torch.empty(v, h, w, dtype=torch.float32, device=dev)
        alpha = torch.empty(v, h, w, dtype=torch.float32, device=dev)
        s0 = GaussianRasterizationSettings(h, w, 0.0, 0.0, None, 1.0, None, None, degree, None, False, False)
        inference = not any(ctx.needs_input_grad[:4])     # no backward will follow: the blend skips the contributor count

        def launch(i, cap_i):   # single-view re-render (capacity overflow)
            dims = R.make_dims(N, shs.shape[2], s0, sh_fp16=shs.dtype == torch.float16, native_layout=True,
                               inference=inference)
            rs, _, _, _ = R._launch_forward(dims, means, cov6, shs, None, opac, bgs[i], views[i], fulls[i],
                                            campos[i], cap_i, tanfov=tanfov[i],
                                            scale=None if scale is None else scale[i],
                                            out=(color[i], depth[i], alpha[i]))
            return rs

        # all v views in ONE library call (fs_raster_forward_views): per-view host work is a few pointer offsets
        # instead of seven allocations and a 23-argument ctypes call, the views alternate over the side streams
        # inside the library and are joined back into the current stream before the call returns
        dims = R.make_dims(N, shs.shape[2], s0, sh_fp16=shs.dtype == torch.float16, native_layout=True, inference=inference)
        n_streams = min(R.NUM_STREAMS, v)
        while len(st.side_streams) < n_streams:
            st.side_streams.append(torch.cuda.Stream(device=dev))
        sz = R._buffer_sizes(N, h, w, cap)
        u8 = lambda n: torch.empty(n, dtype=torch.uint8, device=dev)
        geom, binning, image = u8(v * sz[0]), u8(v * sz[1]), u8(v * sz[2])
        # one key area per view in flight (the projection of a whole batch of views is ONE launch, fs_raster_forward_views)
        scratch = u8(max(R._lib.lib().fs_raster_scratch_slots(v, n_streams if n_streams > 1 else 0), 1) * sz[3])
        radii = torch.empty(v, N, dtype=torch.int32, device=dev)
        counters = torch.empty(v, 2, dtype=torch.int32, device=dev)
        strides = (C.c_size_t * 4)(*sz)
        handles = (C.c_void_p * max(n_streams, 1))(*[s.cuda_stream for s in st.side_streams[:n_streams]])
        p = R._lib.ptr
        if v > 0:
            R._lib.check(R._lib.lib().fs_raster_forward_views(
                C.byref(dims), v, p(means), p(cov6), p(shs), None, p(opac), p(bgs), p(views), p(fulls), p(campos),
                p(tanfov), p(scale), p(geom), p(binning), p(image), p(scratch), strides, cap, p(color), p(depth),
                p(alpha), p(radii), p(counters), n_streams if n_streams > 1 else 0, handles,
                R._lib.current_stream()), "fs_raster_forward_views")
        states = []
        for i in range(v):
            rs = R.RasterState()
            rs.dims = dims
            rs.geom, rs.binning, rs.image = (t[i * n:(i + 1) * n] for t, n in ((geom, sz[0]), (binning, sz[1]), (image, sz[2])))
            rs.radii, rs.counters, rs.cap = radii[i], counters[i], cap
            rs.bg, rs.view, rs.proj, rs.campos = bgs[i], views[i], fulls[i], campos[i]
            rs.tanfov, rs.scale = tanfov[i], None if scale is None else scale[i]
            rs.num_rendered = -1
            states.append(rs)
        # everything the one-call backward needs while the views still live in the strided buffers of this call
        batch = dict(dims=dims, sz=sz, geom=geom, binning=binning, image=image, counters=counters, bgs=bgs, views=views, fulls=fulls,
                     campos=campos, tanfov=tanfov, scale=scale)
        if deferred:
            # without a backward to come only the 8-byte counter pairs stay alive until the check, not the buffers
            _pending_checks.append((counters, states if any(ctx.needs_input_grad) else None, dev, (h, w)))
        else:
            counters = torch.stack([rs.counters for rs in states]).tolist()  # the single sync
            worst, redone = 0, []
            for i, (n_inst, overflow) in enumerate(counters):
                n_inst &= 0xFFFFFFFF
                if overflow:
                    states[i] = launch(i, st.note_overflow(n_inst, overflow & 0xFFFFFFFF, h, w))
                    redone.append(i)
                    batch = None        # that view now lives in its own buffers: backward goes view by view
                states[i].num_rendered = n_inst
                worst = max(worst, n_inst)
            if redone:
                # the relaunches must have fitted: a second overflow (capacity clamp, saturated instance count) would
                # leave the blend unrun and the uninitialised colour buffer returned as the image
                again = torch.stack([states[i].counters for i in redone]).tolist()
                if any(o for _, o in again):
                    raise R._lib.FreeSplatHipError("rasterizer instance list overflowed twice (views "
                                                   f"{[i for i, (_, o) in zip(redone, again) if o]})")
            else:
                st.note_fit(h, w)
            st.last_instances = worst
        ctx.states = states
        ctx.batch = batch
        ctx.save_for_backward(means, cov6, shs, opac)
        ctx.set_materialize_grads(False)
        return color, depth

    @staticmethod
    def backward(ctx, g_color, g_depth):
        means, cov6, shs, opac = ctx.saved_tensors
        if g_color is None and g_depth is None:
            return (None,) * 14
        v = len(ctx.states)
        n_streams = min(R.NUM_STREAMS, v)
        if g_color is not None:
            g_color = g_color.contiguous()
        if g_depth is not None:
            g_depth = g_depth.contiguous()
        dev = means.device
        N = means.shape[0]
        if N == 0 or v == 0:   # nothing rendered: empty / zero gradients (torch hands out NULL pointers for empty tensors)
            return (torch.zeros_like(means), torch.zeros_like(cov6), torch.zeros_like(shs),
                    torch.zeros(N, dtype=torch.float32, device=dev)) + (None,) * 10
        if ctx.batch is not None:
            # one library call: the blend backward of the views alternates over the side streams, then ONE pass
            # over the Gaussians sums the parameter gradients of all views (fs_raster_backward_views)
            b = ctx.batch
            st = R._state(dev)
            while len(st.side_streams) < n_streams:
                st.side_streams.append(torch.cuda.Stream(device=dev))
            if g_color is None:
                g_color = torch.zeros(v, 3, b["dims"].H, b["dims"].W, dtype=torch.float32, device=dev)
            f32 = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
            out = dict(means3D=f32(N, 3), means2D=f32(N, 3), cov3D=f32(*cov6.shape), shs=f32(*shs.shape), opacities=f32(N))
            scratch = torch.empty(v * ((N * 48 + 255) // 256 * 256), dtype=torch.uint8, device=dev)
            strides = (C.c_size_t * 3)(*b["sz"][:3])
            handles = (C.c_void_p * max(n_streams, 1))(*[s.cuda_stream for s in st.side_streams[:n_streams]])
            p = R._lib.ptr
            common = (C.byref(b["dims"]), v, p(means), p(cov6), p(shs), None, p(opac), p(b["bgs"]), p(b["views"]), p(b["fulls"]),
                      p(b["campos"]), p(b["tanfov"]), p(b["scale"]), p(b["geom"]), p(b["binning"]), p(b["image"]), p(b["counters"]),
                      strides, p(g_color), p(g_depth), p(scratch), p(out["means3D"]), p(out["means2D"]), p(out["cov3D"]),
                      p(out["shs"]), None, p(out["opacities"]), 0, n_streams if n_streams > 1 else 0, handles)
            hook = GRAD_EXCHANGE_HOOK
            if hook is None:
                R._lib.check(R._lib.lib().fs_raster_backward_views(*common, R._lib.current_stream()), "fs_raster_backward_views")
            else:
                # chunked gradient exchange (view_sharding.GradExchange("chunked")): the per-Gaussian pass runs chunk by chunk of
                # the rows, and the reduce-scatter of chunk c (on the hook's side stream) overlaps the pass over chunk c + 1
                hook.begin(N)
                for ci, (c0, c1) in enumerate(hook.chunk_rows(N)):
                    R._lib.check(R._lib.lib().fs_raster_backward_views_rows(*common, R._lib.current_stream(), c0, c1 - c0,
                                                                            1 if ci == 0 else 0), "fs_raster_backward_views_rows")
                    hook.chunk_ready(c0, c1, [out["means3D"], out["cov3D"], out["shs"], out["opacities"]])
        else:
            view_grads = lambda i: (None if g_color is None else g_color[i], None if g_depth is None else g_depth[i])
            out = None
            for i, rs in enumerate(ctx.states):
                out = R.rasterize_backward(rs, means, cov6, shs, None, opac, *view_grads(i), out=out, accumulate=i > 0)
            if GRAD_EXCHANGE_HOOK is not None:     # (a re-rendered view forced the view-by-view backward: one chunk, no overlap)
                GRAD_EXCHANGE_HOOK.begin(N)
                GRAD_EXCHANGE_HOOK.chunk_ready(0, N, [out["means3D"], out["cov3D"], out["shs"], out["opacities"]])
        g_shs = out["shs"] if shs.dtype == torch.float32 else out["shs"].to(shs.dtype)
        return (out["means3D"], out["cov3D"], g_shs, out["opacities"]) + (None,) * 10


def render_views(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor,
                 image_shape: tuple[int, int], background_color: Tensor, means: Tensor,
                 covariances: Tensor, sh_coefficients: Tensor, opacities: Tensor,
                 scale_invariant: bool = True, check: str = "now"):
    """v views [v,...] of ONE Gaussian set (means [G,3], covariances [G,3,3], sh [G,3,d_sh],
    opacities [G]).  Same result as render_cuda on v repeated copies to within fp32 rounding of the camera
    matrices (fs_frame_views forms them in double and rounds once; render_cuda, like the reference, chains fp32
    torch ops -- tests/test_raster_hip.py compares the two with a tolerance); the per-view 1/near
    rescale and tan(fov) stay on the device (fs_raster_forward's scale_dev / tanfov_dev), nothing is
    repeated, the images are written straight into the [v,...] outputs, and the only host sync is
    the instance-capacity check: check="now" (default) does it at the end of the call and re-renders
    overflowed views; check="deferred" leaves it to decoder.check_deferred() so that back-to-back
    calls keep the GPU queue full."""
    campos, scale, tanfov, view, full = frame_views(extrinsics, intrinsics, near, far, scale_invariant)
    if not scale_invariant:
        scale = None
    degree = isqrt(sh_coefficients.shape[-1]) - 1
    h, w = image_shape
    # harmonics [G,3,d_sh] and covariances [G,3,3] go to the kernels as they are (FS_RASTER_SH_CHANNEL_MAJOR |
    # FS_RASTER_COV_FULL): no transposed / gathered copies per call (cuda_splatting.py:78, :126)
    color, depth = _RenderViews.apply(means.contiguous(), covariances.contiguous(), sh_coefficients.contiguous(),
                                      opacities.contiguous(), view, full,
                                      campos, background_color.contiguous(), tanfov, scale, h, w, degree,
                                      check == "deferred")
    return color, depth.unsqueeze(1)


@dataclass
class DecoderOutput:
    color: Optional[Tensor]  # [b, v, 3, h, w]
    depth: Optional[Tensor]  # [b, v, h, w]


@dataclass
class Gaussians:
    """src/model/types.py:7-12"""
    means: Tensor        # [b, g, 3]
    covariances: Tensor  # [b, g, 3, 3]
    harmonics: Tensor    # [b, g, 3, d_sh]
    opacities: Tensor    # [b, g]


@dataclass
class DecoderSplattingCUDACfg:
    """decoder_splatting_cuda.py:15-17"""
    name: str = "splatting_cuda"


class DecoderSplattingCUDA(nn.Module):
    """Mirror of decoder_splatting_cuda.py:20-75 (registry key "splatting_cuda"): same constructor
    `(cfg, dataset_cfg)` -- only `dataset_cfg.background_color` is read, as in the reference (:28-32) -- same
    `forward` signature and `DecoderOutput`.

    Beyond the reference (keyword-only, defaults keep the reference's behaviour on one GPU):
      batched=True   render through `render_views` (no v-fold repeat of the Gaussians, one sync per call);
                     False reproduces the reference's repeat-per-view call pattern through `render_cuda`.
      group          a torch.distributed process group (or True for the default group): the v target views of
                     every scene are SHARDED over its ranks (view_sharding.shard_range), each rank renders its
                     block, the images are all-gathered (RCCL over xGMI) so that every rank returns the full
                     [b, v, ...] output, and in backward the per-Gaussian gradients of the shards are summed over
                     the ranks in one flat bucket -- the multi-GPU run of src/main.py:98-103 without replicating
                     the render work.  Gaussians and cameras must be identical on all ranks of the group: the first
                     sharded call (every call with FREESPLAT_CHECK_REPLICAS=1) compares a checksum of the means and
                     the extrinsics across the ranks and raises on a mismatch (Lightning DDP hands every rank its
                     own scene: sharding the views of DIFFERENT scenes would silently mix them).
      single_rank_collectives   take the sharded path (and issue its collectives) even on a one-rank group: how the
                     RCCL branch is exercised on a one-GPU box (tests/test_rccl_world1.py)."""

    def __init__(self, cfg=None, dataset_cfg=None, *, background_color=None, batched: bool = True, group=None,
                 single_rank_collectives: bool = False):
        super().__init__()
        if background_color is None:
            background_color = getattr(dataset_cfg, "background_color", None) if dataset_cfg is not None else None
        if background_color is None:
            if cfg is not None and not hasattr(cfg, "name") and dataset_cfg is None:
                background_color = cfg      # DecoderSplattingCUDA((r, g, b)): round-1 form, kept for callers of it
                cfg = None
            else:
                background_color = (0.0, 0.0, 0.0)
        self.cfg = cfg if cfg is not None else DecoderSplattingCUDACfg()
        self.dataset_cfg = dataset_cfg
        self.register_buffer("background_color", torch.tensor(list(background_color), dtype=torch.float32),
                             persistent=False)
        self.batched = batched
        self.group = group
        self.single_rank_collectives = single_rank_collectives
        self._replicas_checked = None      # Gaussian count of the last scene whose replicas were compared
        self._sharded_calls = 0            # sharded forward calls so far (the check is also repeated every N-th call)

    def _dist_group(self):
        if self.group is None or self.group is False:
            return None
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("DecoderSplattingCUDA(group=...) needs an initialised torch.distributed process group")
        g = None if self.group is True else self.group      # None = the default group
        return (g, dist) if (dist.get_world_size(g) > 1 or self.single_rank_collectives) else None

    def _check_replicas(self, group, dist, gaussians, extrinsics):
        """All ranks of the group must hold the same scene: compare a cheap signature (Gaussian count; sum of magnitudes and
        sum of squares of the means and of the extrinsics) through one MIN and one MAX all-reduce of 5 doubles.  The count must
        be equal; the sums within 1e-6 relative -- replicated encoders agree to rounding, not to the bit (MIOpen algorithm
        choice, float atomics in the encoder's backward kernels), while different scenes differ in the first digits.  Every
        entry is a sum of NON-NEGATIVE terms: a signed sum that cancels towards zero (a centred scene) would turn legitimate
        rounding differences into a relative error above any tolerance."""
        from .view_sharding import _stage
        m, e = gaussians.means.detach().double(), extrinsics.detach().double()
        sig = torch.stack([torch.tensor(float(m.shape[-2]), dtype=torch.float64, device=m.device), m.abs().sum(), (m * m).sum(),
                           e.abs().sum(), (e * e).sum()])
        lo, hi = _stage(sig.clone(), group), _stage(sig.clone(), group)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
        if lo[0] != hi[0] or bool(((hi - lo).abs() > 1e-6 * torch.maximum(hi.abs(), lo.abs()) + 1e-12).any()):
            raise RuntimeError("DecoderSplattingCUDA(group=...): Gaussians / cameras differ between the ranks of the group "
                               f"(checksum min {lo.tolist()} != max {hi.tolist()}); view sharding needs the SAME scene on "
                               "every rank (shard scenes over ranks with group=None instead)")

    def forward(self, gaussians, extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor,
                image_shape: tuple[int, int], depth_mode=None, no_color: bool = False) -> DecoderOutput:
        b, v = extrinsics.shape[:2]
        if no_color:
            if depth_mode is not None:
                # reference raises UnboundLocalError here (decoder_splatting_cuda.py:47-71)
                raise RuntimeError("no_color=True with depth_mode set has no defined result in the reference")
            return DecoderOutput(None, None)
        bg = self.background_color
        sharded = self._dist_group()
        if sharded is not None:
            # on the first sharded call, whenever the Gaussian count changes (a new scene), every
            # FREESPLAT_CHECK_REPLICAS_EVERY-th call (default 64: scenes of EQUAL count -- fixed-size outputs without PTF --
            # are re-validated too), and always with FREESPLAT_CHECK_REPLICAS=1
            n_now = int(gaussians.means.shape[-2])
            every = max(1, int(os.environ.get("FREESPLAT_CHECK_REPLICAS_EVERY", "64")))
            if (self._replicas_checked != n_now or self._sharded_calls % every == 0
                    or os.environ.get("FREESPLAT_CHECK_REPLICAS") == "1"):
                self._check_replicas(sharded[0], sharded[1], gaussians, extrinsics)
                self._replicas_checked = n_now
            self._sharded_calls += 1
            color, depth = self._forward_sharded(sharded[0], sharded[1], gaussians, extrinsics, intrinsics, near, far,
                                                 image_shape, with_depth=depth_mode is not None)
            if depth is None:
                return DecoderOutput(color, None)
        elif self.batched:
            colors, depths = [], []
            for i in range(b):
                c, d = render_views(extrinsics[i], intrinsics[i], near[i], far[i], image_shape,
                                    bg[None].expand(v, 3), gaussians.means[i], gaussians.covariances[i],
                                    gaussians.harmonics[i], gaussians.opacities[i])
                colors.append(c); depths.append(d)
            color = torch.stack(colors)
            depth = torch.stack(depths).squeeze(2)
        else:
            rep = lambda t: t[:, None].expand(b, v, *t.shape[1:]).reshape(b * v, *t.shape[1:])
            color, depth = render_cuda(
                extrinsics.reshape(b * v, 4, 4), intrinsics.reshape(b * v, 3, 3), near.reshape(b * v),
                far.reshape(b * v), image_shape, bg[None].expand(b * v, 3), rep(gaussians.means),
                rep(gaussians.covariances), rep(gaussians.harmonics), rep(gaussians.opacities))
            color = color.reshape(b, v, *color.shape[1:])
            depth = depth.reshape(b, v, *depth.shape[1:]).squeeze(2)
        depth = depth / 2  # decoder_splatting_cuda.py:62
        return DecoderOutput(color, None if depth_mode is None else depth)

    def render_depth(self, gaussians, extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor,
                     image_shape: tuple[int, int], mode: str = "depth") -> Tensor:
        """decoder_splatting_cuda.py:77-100: [b,v,...] cameras, one Gaussian set per scene -> [b,v,h,w]."""
        b, v = extrinsics.shape[:2]
        rep = lambda t: t[:, None].expand(b, v, *t.shape[1:]).reshape(b * v, *t.shape[1:])
        result = render_depth_cuda(extrinsics.reshape(b * v, 4, 4), intrinsics.reshape(b * v, 3, 3), near.reshape(b * v),
                                   far.reshape(b * v), image_shape, rep(gaussians.means), rep(gaussians.covariances),
                                   rep(gaussians.opacities), mode=mode)
        return result.reshape(b, v, *result.shape[1:])

    def _forward_sharded(self, group, dist, gaussians, extrinsics, intrinsics, near, far, image_shape, with_depth=True):
        """View-sharded rendering of every scene of the batch: no collective on the render
        path itself; one all-gather of colour (+depth only when the caller asked for it: 15 instead of 20 MB per
        968x1296 view) per scene, one flat-bucket gradient sum in backward."""
        from .view_sharding import gather_views_autograd, replicate_gaussians, shard_range
        b, v = extrinsics.shape[:2]
        h, w = image_shape
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        mine = shard_range(v, rank, world)
        sl = slice(mine.start, mine.stop)
        bg = self.background_color
        colors, depths = [], []
        for i in range(b):
            means, cov, sh, op = replicate_gaussians([gaussians.means[i], gaussians.covariances[i],
                                                      gaussians.harmonics[i], gaussians.opacities[i]], group)
            if len(mine):
                c, d = render_views(extrinsics[i, sl], intrinsics[i, sl], near[i, sl], far[i, sl], image_shape,
                                    bg[None].expand(len(mine), 3), means, cov, sh, op)
                local = torch.cat([c, d], dim=1) if with_depth else c   # [v_local, 4 | 3, h, w]
            else:   # more ranks than views: this rank contributes nothing (but must keep the graph connected)
                local = (torch.zeros(0, 4 if with_depth else 3, h, w, device=extrinsics.device)
                         + 0.0 * (means.sum() + cov.sum() + sh.sum() + op.sum()))
            full = gather_views_autograd(local, v, group)              # [v, 4 | 3, h, w] on every rank
            colors.append(full[:, :3])
            if with_depth:
                depths.append(full[:, 3])
        return torch.stack(colors), (torch.stack(depths) if with_depth else None)
