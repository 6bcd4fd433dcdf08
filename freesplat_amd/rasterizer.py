"""Host side of the MI355X rasterizer behind the reference's operator API.

Mirrors the interface FreeSplat imports from the (un-vendored) CUDA extension
`diff_gaussian_rasterization_depth` at /root/reference/src/model/decoder/cuda_splatting.py:5-8
and calls at :100-127:

    settings   = GaussianRasterizationSettings(image_height, image_width, tanfovx, tanfovy, bg,
                     scale_modifier, viewmatrix, projmatrix, sh_degree, campos, prefiltered, debug)
    rasterizer = GaussianRasterizer(settings)
    color, radii, depth, alpha = rasterizer(means3D, means2D, shs=..., colors_precomp=...,
                                            opacities=..., cov3D_precomp=...)

Same names, argument meaning and error behaviour (Python exceptions).  The compute is
libfreesplat_hip.so (C ABI, include/freesplat_amd.h) on the current HIP stream; there is no CPU
or eager fallback -- a CPU tensor raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import NamedTuple, Optional

import torch
from torch import Tensor, nn

from . import _lib


# Precise tile culling (include/freesplat_amd.h FS_RASTER_TILE_CULL): images unchanged bit for bit,
# tile lists shrink to the instances that can actually contribute.  FREESPLAT_TILE_CULL=0 keeps the
# reference's full 3-sigma-square lists.
TILE_CULL = os.environ.get("FREESPLAT_TILE_CULL", "1") != "0"
# Blend-loop exponential.  Default: the CPU-reproducible polynomial exp of the bit-exact contract (forward identical to
# the oracle bit for bit).  FREESPLAT_FAST_EXP=1 (or rasterizer.FAST_EXP = True) selects the hardware v_exp_f32
# (include/freesplat_amd.h FS_RASTER_FAST_EXP): +7 % views/s at config 3; lists / ordering / radii stay identical, the
# alpha >= 1/255 decisions are those of the exact mode (guard band: a step with an alpha within 16 ulp of the threshold
# is re-evaluated with the contract exp) and the image stays within ~1e-6 of the contract.  What is left is the
# T >= 1e-4 termination: a flipped one changes a pixel by less than its remaining transmittance, 1e-4 x colour (about one
# pixel per 2 M; counted in tests and in bench.py's `fast_exp` block).  Opt-in because it is not BIT-exact.
FAST_EXP = os.environ.get("FREESPLAT_FAST_EXP", "0") == "1"
# Views of one render_views call are spread round-robin over this many HIP streams so that the short
# latency-bound launches of one view (tile scan, kernel tails) overlap the VALU-bound blend of another.
NUM_STREAMS = max(1, int(os.environ.get("FREESPLAT_RASTER_STREAMS", "2")))  # views of one call round-robin over this many streams


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: Tensor
    scale_modifier: float
    viewmatrix: Tensor
    projmatrix: Tensor
    sh_degree: int
    campos: Tensor
    prefiltered: bool
    debug: bool


# ---------------------------------------------------------------------------------------------
# workspace management (plumbing): the C ABI never allocates, torch's caching allocator does.
# ---------------------------------------------------------------------------------------------
class _DeviceState:
    """Per-device scratch (reused across calls on the same stream) and capacity history."""

    def __init__(self):
        self.scratch: dict[int, Tensor] = {}   # one reusable scratch per HIP stream
        self.last_instances = 0
        # capacity a past overflow asked for (retry_capacity), per image size; decays on every call that fits (note_fit):
        # one close-up view must not inflate the scratch of every later call for good
        self.retry_caps: dict[tuple[int, int], int] = {}
        self.side_streams: list = []

    @property
    def retry_cap(self) -> int:                # (largest live value: bench / tests reset it through the setter)
        return max(self.retry_caps.values(), default=0)

    @retry_cap.setter
    def retry_cap(self, v: int) -> None:
        if v == 0:
            self.retry_caps.clear()
        else:
            raise ValueError("retry_cap can only be reset to 0; use note_overflow")

    def note_overflow(self, n_inst: int, max_tile: int, H: int, W: int) -> int:
        cap = max(self.retry_caps.get((H, W), 0), retry_capacity(n_inst, max_tile, H, W))
        self.retry_caps[(H, W)] = cap
        return cap

    def note_fit(self, H: int, W: int) -> None:
        """A call of this image size fitted its capacity: let the overflow-derived capacity decay (x0.9 per call; below the
        default capacity it no longer matters and is dropped)."""
        c = self.retry_caps.get((H, W))
        if c is not None:
            c = int(c * 0.9)
            if c < (1 << 20):
                del self.retry_caps[(H, W)]
            else:
                self.retry_caps[(H, W)] = c


_states: dict[int, _DeviceState] = {}


def _state(device: torch.device) -> _DeviceState:
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _states.get(idx)
    if st is None:
        st = _states[idx] = _DeviceState()
    return st


def _buffer_sizes(N: int, H: int, W: int, cap: int):
    out = (C.c_size_t * 4)()
    _lib.check(_lib.lib().fs_raster_buffer_sizes(N, H, W, cap, out), "fs_raster_buffer_sizes")
    return [int(x) for x in out]


def default_capacity(N: int, st: _DeviceState, H: int = 0, W: int = 0) -> int:
    """Instance capacity: generous (HBM is 288 GB) so the overflow retry is the rare path."""
    return max(1 << 20, 8 * N, int(st.last_instances * 1.25) + 1024, st.retry_caps.get((H, W), 0))


def retry_capacity(n_inst: int, max_tile: int, H: int, W: int) -> int:
    """Capacity for the retry after an overflow: counters = {instances, largest tile list}.  The saved lists need
    `n_inst` entries; every tile owns a fixed key area of tile_capacity(cap, T) = pow2 >= 4*cap/T slots
    (csrc/fs_common.h), which must hold the largest tile list."""
    T = ((W + 15) // 16) * ((H + 15) // 16)
    return max(n_inst + 1024, (max_tile * T + 3) // 4 + 1)


class RasterState:
    """Everything one forward leaves behind for its backward (device buffers + dims)."""

    __slots__ = ("dims", "geom", "binning", "image", "radii", "counters", "cap", "bg", "view", "proj",
                 "campos", "tanfov", "scale", "num_rendered")


def _launch_forward(dims: _lib.RasterDims, means3D, cov3D, shs, colors, opacities, bg, view, proj,
                    campos, cap: int, tanfov: Optional[Tensor] = None, scale: Optional[Tensor] = None,
                    out: Optional[tuple] = None) -> tuple[RasterState, Tensor, Tensor, Tensor]:
    """One stream-ordered forward launch (no host sync).  `tanfov` [2] / `scale` [1]: optional
    device-resident settings; `out` = (color[3,H,W], depth[H,W], alpha[H,W]) views to write into."""
    dev = means3D.device
    N, H, W = dims.N, dims.H, dims.W
    sz = _buffer_sizes(N, H, W, cap)
    st = _state(dev)
    skey = torch.cuda.current_stream().cuda_stream
    scratch = st.scratch.get(skey)
    if scratch is None or scratch.numel() < sz[3]:
        scratch = st.scratch[skey] = torch.empty(sz[3], dtype=torch.uint8, device=dev)
    rs = RasterState()
    rs.dims = dims
    rs.geom = torch.empty(sz[0], dtype=torch.uint8, device=dev)
    rs.binning = torch.empty(sz[1], dtype=torch.uint8, device=dev)
    rs.image = torch.empty(sz[2], dtype=torch.uint8, device=dev)
    rs.radii = torch.empty(N, dtype=torch.int32, device=dev)
    rs.counters = torch.empty(2, dtype=torch.int32, device=dev)
    rs.cap = cap
    rs.bg, rs.view, rs.proj, rs.campos = bg, view, proj, campos
    rs.tanfov, rs.scale = tanfov, scale
    rs.num_rendered = -1
    if out is None:
        color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
        depth = torch.empty(H, W, dtype=torch.float32, device=dev)
        alpha = torch.empty(H, W, dtype=torch.float32, device=dev)
    else:
        color, depth, alpha = out
    p = _lib.ptr
    _lib.check(_lib.lib().fs_raster_forward(
        C.byref(dims), p(means3D), p(cov3D), p(shs), p(colors), p(opacities), p(bg), p(view), p(proj),
        p(campos), p(tanfov), p(scale), p(rs.geom), p(rs.binning), p(rs.image), p(scratch), cap,
        p(color), p(depth),
        p(alpha), p(rs.radii), p(rs.counters), _lib.current_stream()), "fs_raster_forward")
    return rs, color, depth, alpha


def _f32c(t: Tensor, name: str, allow_half: bool = False) -> Tensor:
    if t.device.type != "cuda":
        raise RuntimeError(f"freesplat_amd rasterizer: `{name}` must live on a HIP device "
                           f"(got {t.device}); there is no CPU path")
    if allow_half and t.dtype == torch.float16:
        return t.contiguous()
    if t.dtype != torch.float32:
        raise RuntimeError(f"freesplat_amd rasterizer: `{name}` must be float32 (got {t.dtype})")
    return t.contiguous()


def make_dims(N, M, settings: GaussianRasterizationSettings, sh_fp16: bool = False, native_layout: bool = False,
              inference: bool = False) -> _lib.RasterDims:
    """`native_layout`: shs is the reference's harmonics [N,3,M] and cov3D its covariances [N,3,3]
    (FS_RASTER_SH_CHANNEL_MAJOR | FS_RASTER_COV_FULL) instead of the rasterizer API's [N,M,3] / [N,6].
    `inference`: no backward will follow (FS_RASTER_NO_BACKWARD_STATE: the blend skips the contributor count)."""
    d = _lib.RasterDims()
    d.N, d.M = int(N), int(M)
    d.H, d.W = int(settings.image_height), int(settings.image_width)
    d.sh_degree = int(settings.sh_degree)
    d.tanfovx, d.tanfovy = float(settings.tanfovx), float(settings.tanfovy)
    d.flags = ((_lib.RASTER_TILE_CULL if TILE_CULL else 0) | (_lib.RASTER_SH_FP16 if sh_fp16 else 0)
               | (_lib.RASTER_FAST_EXP if FAST_EXP else 0) | (_lib.RASTER_NO_BACKWARD_STATE if inference else 0))
    if native_layout:
        d.flags |= _lib.RASTER_SH_CHANNEL_MAJOR | _lib.RASTER_COV_FULL
    return d


def rasterize_forward_checked(dims, means3D, cov3D, shs, colors, opacities, bg, view, proj, campos):
    """Forward + capacity check (one host sync on the 8-byte counter pair), retrying once with the
    exact capacity when the instance list overflowed."""
    st = _state(means3D.device)
    cap = default_capacity(dims.N, st, dims.H, dims.W)
    rs, color, depth, alpha = _launch_forward(dims, means3D, cov3D, shs, colors, opacities, bg, view,
                                              proj, campos, cap)
    n_inst, overflow = (int(x) & 0xFFFFFFFF for x in rs.counters.tolist())
    if not overflow:
        st.note_fit(dims.H, dims.W)
    else:          # (non-zero = the largest tile list)
        cap = st.note_overflow(n_inst, overflow, dims.H, dims.W)
        rs, color, depth, alpha = _launch_forward(dims, means3D, cov3D, shs, colors, opacities, bg,
                                                  view, proj, campos, cap)
        n_inst, overflow = (int(x) & 0xFFFFFFFF for x in rs.counters.tolist())
        if overflow:
            raise _lib.FreeSplatHipError("rasterizer instance list overflowed twice")
    rs.num_rendered = n_inst
    st.last_instances = n_inst
    return rs, color, depth, alpha


def rasterize_backward(rs: RasterState, means3D, cov3D, shs, colors, opacities, g_color, g_depth, out=None,
                       accumulate: bool = False):
    """Launch the backward of one view.  `out` = dict of preallocated gradient tensors (for the
    multi-view accumulate path) or None to allocate."""
    dev = means3D.device
    d = rs.dims
    N = d.N
    if out is None:
        out = dict(
            means3D=torch.empty(N, 3, dtype=torch.float32, device=dev),
            means2D=torch.empty(N, 3, dtype=torch.float32, device=dev),
            cov3D=torch.empty(cov3D.shape, dtype=torch.float32, device=dev),
            shs=None if shs is None else torch.empty(shs.shape, dtype=torch.float32, device=dev),
            colors=None if colors is None else torch.empty(N, 3, dtype=torch.float32, device=dev),
            opacities=torch.empty(N, dtype=torch.float32, device=dev),
        )
    scratch = torch.empty(max(N, 1) * 12, dtype=torch.float32, device=dev)
    if g_color is None:
        g_color = torch.zeros(3, d.H, d.W, dtype=torch.float32, device=dev)
    g_color = g_color.contiguous()
    if g_depth is not None:
        g_depth = g_depth.contiguous()
    p = _lib.ptr
    _lib.check(_lib.lib().fs_raster_backward(
        C.byref(d), p(means3D), p(cov3D), p(shs), p(colors), p(opacities), p(rs.bg), p(rs.view), p(rs.proj),
        p(rs.campos), p(rs.tanfov), p(rs.scale), p(rs.geom), p(rs.binning), p(rs.image), p(rs.counters), p(g_color),
        p(g_depth), p(scratch),
        p(out["means3D"]), p(out["means2D"]), p(out["cov3D"]), p(out["shs"]), p(out["colors"]),
        p(out["opacities"]), 1 if accumulate else 0, _lib.current_stream()), "fs_raster_backward")
    return out


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, shs, colors_precomp, opacities, cov3D, settings):
        N = means3D.shape[0]
        M = 0 if shs is None else shs.shape[1]
        dims = make_dims(N, M, settings, sh_fp16=shs is not None and shs.dtype == torch.float16,
                         inference=not any(ctx.needs_input_grad[:6]))
        bg = _f32c(settings.bg, "bg")
        view = _f32c(settings.viewmatrix, "viewmatrix")
        proj = _f32c(settings.projmatrix, "projmatrix")
        campos = _f32c(settings.campos, "campos")
        rs, color, depth, alpha = rasterize_forward_checked(dims, means3D, cov3D, shs, colors_precomp,
                                                            opacities, bg, view, proj, campos)
        ctx.rs = rs
        ctx.opac_shape = None
        ctx.save_for_backward(means3D, cov3D, shs, colors_precomp, opacities)
        ctx.mark_non_differentiable(rs.radii)
        ctx.set_materialize_grads(False)
        return color, rs.radii, depth, alpha

    @staticmethod
    def backward(ctx, g_color, _g_radii, g_depth, g_alpha):
        means3D, cov3D, shs, colors, opacities = ctx.saved_tensors
        if g_alpha is not None:
            raise NotImplementedError("gradient through the accumulated-alpha output is not supported "
                                      "(no reference caller uses it; cuda_splatting.py:120)")
        if g_color is None and g_depth is None:
            return (None,) * 7
        g = rasterize_backward(ctx.rs, means3D, cov3D, shs, colors, opacities, g_color, g_depth)
        g_shs = g["shs"] if (shs is None or shs.dtype == torch.float32) else g["shs"].to(shs.dtype)
        return g["means3D"], g["means2D"], g_shs, g["colors"], g["opacities"], g["cov3D"], None


def build_cov3d(scales: Tensor, rotations: Tensor, scale_modifier: float) -> Tensor:
    """cov3D 6-vector from scales [N,3] and rotations [N,4] (w,x,y,z) -- the (scales, rotations)
    input form of the original extension; plain differentiable torch ops (host-side plumbing, not
    on FreeSplat's path, which always passes cov3D_precomp: cuda_splatting.py:126)."""
    q = rotations / rotations.norm(dim=-1, keepdim=True)
    r, x, y, z = q.unbind(-1)
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
    Mx = R * (scales * scale_modifier)[:, None, :]
    S = Mx @ Mx.transpose(1, 2)
    i, j = torch.triu_indices(3, 3)
    return S[:, i, j]


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None,
                rotations=None, cov3D_precomp=None):
        s = self.raster_settings
        if (shs is None) == (colors_precomp is None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        means3D = _f32c(means3D, "means3D")
        N = means3D.shape[0]
        if cov3D_precomp is None:
            cov3D_precomp = build_cov3d(scales, rotations, float(s.scale_modifier))
        cov3D = _f32c(cov3D_precomp, "cov3D_precomp").reshape(N, 6)
        opac = _f32c(opacities, "opacities").reshape(N)
        if shs is not None:
            shs = _f32c(shs, "shs", allow_half=True)   # fp16 = storage-only SH (FS_RASTER_SH_FP16)
            if shs.dim() != 3 or shs.shape[0] != N or shs.shape[2] != 3:
                raise RuntimeError(f"shs must be [N, M, 3], got {tuple(shs.shape)}")
            if (int(s.sh_degree) + 1) ** 2 > shs.shape[1] or int(s.sh_degree) > 3:
                raise RuntimeError(f"sh_degree {s.sh_degree} unsupported for shs with {shs.shape[1]} coefficients "
                                   "(degree <= 3)")
        else:
            colors_precomp = _f32c(colors_precomp, "colors_precomp").reshape(N, 3)
        if means2D is None:
            means2D = torch.zeros(N, 3, dtype=torch.float32, device=means3D.device)
        return _RasterizeGaussians.apply(means3D, means2D, shs, colors_precomp, opac, cov3D, s)


# --- debug views into the opaque buffers (tests only) -------------------------------------------
def debug_state(rs: RasterState) -> dict:
    """Copy the forward's internal state to host numpy arrays (test helper)."""
    import numpy as np
    d = rs.dims
    T = ((d.W + 15) // 16) * ((d.H + 15) // 16)
    off_bytes = ((T + 1) * 4 + 255) // 256 * 256
    binning = rs.binning.cpu().numpy()
    offsets = binning[: (T + 1) * 4].view(np.uint32).copy()
    I = int(offsets[-1])
    words = binning[off_bytes: off_bytes + I * 4].view(np.uint32).copy()
    point_list, quad = words >> 4, words & 15
    geom = rs.geom.cpu().numpy()
    N = d.N
    rec = geom[: N * 48].view(np.float32).reshape(N, 12).copy()
    o1 = (N * 48 + 255) // 256 * 256
    rect = geom[o1: o1 + N * 8].view(np.uint16).reshape(N, 4).copy()
    o2 = o1 + (N * 8 + 255) // 256 * 256
    clamp = geom[o2: o2 + N].copy()
    P = d.H * d.W
    img = rs.image.cpu().numpy()
    final_T = img[: P * 4].view(np.float32).reshape(d.H, d.W).copy()
    o3 = (P * 4 + 255) // 256 * 256
    n_contrib = img[o3: o3 + P * 4].view(np.int32).reshape(d.H, d.W).copy()
    return dict(offsets=offsets, point_list=point_list, quad=quad, rec=rec, rect=rect, clamp=clamp,
                final_T=final_T, n_contrib=n_contrib, radii=rs.radii.cpu().numpy(),
                num_rendered=I)
