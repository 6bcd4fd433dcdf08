"""Drop-in for the reference's GaussianAdapter on FreeSplat's fused path.

Mirrors /root/reference/src/model/encoder/common/gaussian_adapter.py:
    GaussianAdapter(cfg).forward(extrinsics, intrinsics, coordinates, depths, opacities, raw_gaussians,
                                 image_shape, eps=1e-8, fusion=False, coords=None)        (:135-201)
with the same argument shapes as encoder_freesplat.py:317-326 (fusion=True -> world means) and
:376-386 (fusion=False with `coords` -> Gaussians), `d_sh` / `d_in`, the non-persistent `sh_mask`
buffer and `get_scale_multiplier`.  Compute = fs_unproject_* / fs_gaussian_head_* in
libfreesplat_hip.so (forward and backward kernels; differentiable w.r.t. depths, raw channels and
the blended extrinsics).  The pixelSplat-only branch (fusion=False without coords: per-ray means +
SH rotation) is not on FreeSplat's path and raises.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch
from torch import Tensor, nn

from . import _lib


@dataclass
class Gaussians:
    means: Tensor
    covariances: Tensor
    scales: Tensor
    rotations: Tensor
    harmonics: Tensor
    opacities: Tensor


@dataclass
class GaussianAdapterCfg:
    gaussian_scale_min: float
    gaussian_scale_max: float
    sh_degree: int
    load_depth: bool = False


def _chk(t: Tensor, name: str) -> Tensor:
    if t.device.type != "cuda":
        raise RuntimeError(f"freesplat_amd adapter: `{name}` must live on a HIP device (got {t.device}); no CPU path")
    return t.float().contiguous()


class _Unproject(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depths, extrinsics, k0, h, w):
        V = depths.shape[0]
        xyz = torch.empty(V, h * w, 3, dtype=torch.float32, device=depths.device)
        p = _lib.ptr
        _lib.check(_lib.lib().fs_unproject_forward(V, h, w, p(depths), p(extrinsics), p(k0), p(xyz),
                                                   _lib.current_stream()), "fs_unproject_forward")
        ctx.save_for_backward(extrinsics, k0)
        ctx.hw = (V, h, w)
        return xyz

    @staticmethod
    def backward(ctx, g):
        extrinsics, k0 = ctx.saved_tensors
        V, h, w = ctx.hw
        gd = torch.empty(V, h * w, dtype=torch.float32, device=g.device)
        p = _lib.ptr
        _lib.check(_lib.lib().fs_unproject_backward(V, h, w, p(extrinsics), p(k0), p(g.contiguous()), p(gd),
                                                    _lib.current_stream()), "fs_unproject_backward")
        return gd, None, None, None, None


class _Head(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, depths, extrinsics, mult, sh_mask, smin, smax):
        M = raw.shape[0]
        dev = raw.device
        cov = torch.empty(M, 3, 3, device=dev)
        sh = torch.empty(M, 3, 9, device=dev)
        scales = torch.empty(M, 3, device=dev)
        rot = torch.empty(M, 4, device=dev)
        stride = 0 if mult.numel() == 1 else 1
        p = _lib.ptr
        _lib.check(_lib.lib().fs_gaussian_head_forward(M, p(raw), p(depths), p(extrinsics), p(mult), stride, p(sh_mask),
                                                       C.c_float(smin), C.c_float(smax), p(cov), p(sh), p(scales), p(rot),
                                                       _lib.current_stream()), "fs_gaussian_head_forward")
        ctx.save_for_backward(raw, depths, extrinsics, mult, sh_mask)
        ctx.cfg = (smin, smax, stride)
        ctx.set_materialize_grads(False)
        return cov, sh, scales, rot

    @staticmethod
    def backward(ctx, g_cov, g_sh, g_scales, g_rot):
        raw, depths, extrinsics, mult, sh_mask = ctx.saved_tensors
        smin, smax, stride = ctx.cfg
        M = raw.shape[0]
        g_raw, g_dep, g_E = torch.empty_like(raw), torch.empty_like(depths), torch.empty_like(extrinsics)
        c = lambda t: None if t is None else t.contiguous()
        p = _lib.ptr
        gc_, gs_, gsc_, gr_ = c(g_cov), c(g_sh), c(g_scales), c(g_rot)   # (kept alive until the launch is queued)
        _lib.check(_lib.lib().fs_gaussian_head_backward(M, p(raw), p(depths), p(extrinsics), p(mult), stride, p(sh_mask),
                                                        C.c_float(smin), C.c_float(smax), p(gc_), p(gs_), p(gsc_), p(gr_),
                                                        p(g_raw), p(g_dep), p(g_E), _lib.current_stream()),
                   "fs_gaussian_head_backward")
        return g_raw, g_dep, g_E, None, None, None, None


class _LatentsPack(torch.autograd.Function):
    """(head [N,65,h,w], skip [N,64,h,w]) -> (latents [N, h*w, 64] = head[:, 1:] + skip in the fold's pixel-major layout,
    dens [N, h*w] = head[:, 0]): fs_latents_pack_forward / _backward."""

    @staticmethod
    def forward(ctx, head, skip):
        N, P = head.shape[0], head.shape[2] * head.shape[3]
        lat = torch.empty(N, P, 64, dtype=torch.float32, device=head.device)
        dens = torch.empty(N, P, dtype=torch.float32, device=head.device)
        p = _lib.ptr
        _lib.check(_lib.lib().fs_latents_pack_forward(N, P, 64, p(head), p(skip), p(lat), p(dens), _lib.current_stream()),
                   "fs_latents_pack_forward")
        ctx.shapes = (head.shape, skip.shape)
        return lat, dens

    @staticmethod
    def backward(ctx, g_lat, g_dens):
        hs, ss = ctx.shapes
        need_h, need_s = ctx.needs_input_grad
        if (g_lat is None and g_dens is None) or not (need_h or need_s):
            return None, None
        dev = g_lat.device if g_lat is not None else g_dens.device
        g_head = torch.empty(hs, dtype=torch.float32, device=dev) if need_h else None
        g_skip = torch.empty(ss, dtype=torch.float32, device=dev) if need_s else None
        gl = None if g_lat is None else g_lat.float().contiguous()
        gd = None if g_dens is None else g_dens.float().contiguous()
        p = _lib.ptr
        _lib.check(_lib.lib().fs_latents_pack_backward(hs[0], hs[2] * hs[3], 64, p(gl), p(gd), p(g_head), p(g_skip),
                                                       _lib.current_stream()), "fs_latents_pack_backward")
        return g_head, g_skip


def latents_pack(head: Tensor, skip: Tensor):
    """The per-pixel latents and density logits of encoder_freesplat.py:311-316 from the depth decoder's head map
    [(b v), 1 + 64, h, w] and the skip convolution's output [(b v), 64, h, w]:
        latents [(b v), h*w, 64] = rearrange(head[:, 1:] + skip, "n c h w -> n (h w) c"),   dens [(b v), h*w] = head[:, 0]
    -- one HIP pass each way (csrc/adapter.hip) instead of torch's add plus a transposing copy forward and two backward."""
    if head.device.type != "cuda":
        raise RuntimeError(f"freesplat_amd latents_pack: tensors must live on a HIP device (got {head.device}); no CPU path")
    if head.dim() != 4 or skip.dim() != 4 or head.shape[1] != 65 or skip.shape[1] != 64 or head.shape[0] != skip.shape[0] \
            or head.shape[2:] != skip.shape[2:]:
        raise RuntimeError(f"latents_pack: head {tuple(head.shape)} / skip {tuple(skip.shape)}: expected [N,65,h,w] and [N,64,h,w]")
    return _LatentsPack.apply(_chk(head, "head"), _chk(skip, "skip"))


class GaussianAdapter(nn.Module):
    def __init__(self, cfg: GaussianAdapterCfg):
        super().__init__()
        self.cfg = cfg
        self.register_buffer("sh_mask", torch.ones((self.d_sh,), dtype=torch.float32), persistent=False)
        for degree in range(1, self.cfg.sh_degree + 1):
            self.sh_mask[degree ** 2: (degree + 1) ** 2] = 0.1 * 0.25 ** degree

    @property
    def d_sh(self) -> int:
        return (self.cfg.sh_degree + 1) ** 2

    @property
    def d_in(self) -> int:
        return 7 + 3 * self.d_sh

    def get_scale_multiplier(self, intrinsics: Tensor, pixel_size: Tensor, multiplier: float = 0.1) -> Tensor:
        """gaussian_adapter.py:203-214.  Broadcast batch dimensions (the encoder hands in ONE camera `expand()`-ed over the M
        fused Gaussians, encoder_freesplat.py:378) are collapsed to size 1 first: inverting the same 2x2 matrix M = 1.6 M times
        was 3.5 ms of rocsolver kernels per config-3 training step (profiles/r5_c3_step_glue.json).  The result broadcasts
        against the leading shape exactly as before."""
        K = intrinsics[..., :2, :2]
        lead = K.shape[:-2]
        K = K[tuple(slice(0, 1) if (st == 0 and n > 1) else slice(None) for n, st in zip(lead, K.stride()[:-2]))]
        xy = multiplier * torch.einsum("...ij,j->...i", torch.linalg.inv_ex(K).inverse, pixel_size)
        return xy.sum(dim=-1)

    def forward(self, extrinsics, intrinsics, coordinates, depths, opacities, raw_gaussians, image_shape,
                eps: float = 1e-8, fusion: bool = False, coords=None):
        h, w = image_shape
        if fusion:
            # extrinsics [b,v,1,1,1,4,4], intrinsics [b,v,1,1,1,3,3], depths [b,v,h*w,1,1] -> [b,v,h*w,1,1,3]
            b, v = intrinsics.shape[:2]
            out = []
            for i in range(b):
                K = intrinsics[i, 0].reshape(3, 3)
                k0 = torch.stack([K[0, 0] * w, K[1, 1] * h, K[0, 2] * w, K[1, 2] * h])
                xyz = _Unproject.apply(_chk(depths[i].reshape(v, h * w), "depths"),
                                       _chk(extrinsics[i].reshape(v, 4, 4), "extrinsics"), _chk(k0, "intrinsics"), h, w)
                out.append(xyz)
            return torch.stack(out)[:, :, :, None, None, :]
        if coords is None:
            raise NotImplementedError("GaussianAdapter.forward(fusion=False, coords=None) is pixelSplat's per-ray path "
                                      "(get_world_rays + rotate_sh); FreeSplat never takes it "
                                      "(gaussian_adapter.py:174-192)")
        if self.d_sh != 9:
            raise NotImplementedError("the HIP Gaussian head implements sh_degree 2 (d_in = 34)")
        lead = opacities.shape
        M = opacities.numel()
        pixel_size = 1 / torch.tensor((w, h), dtype=torch.float32, device=extrinsics.device)
        mult = self.get_scale_multiplier(intrinsics, pixel_size)
        mult = mult.expand(lead).reshape(M) if mult.numel() > 1 else mult.reshape(1)
        raw = raw_gaussians.expand(*lead, raw_gaussians.shape[-1]).reshape(M, -1)
        E = extrinsics.expand(*lead, 4, 4).reshape(M, 4, 4)
        cov, sh, scales, rot = _Head.apply(_chk(raw, "raw_gaussians"), _chk(depths.expand(lead).reshape(M), "depths"),
                                           _chk(E, "extrinsics"), _chk(mult, "multiplier"), self.sh_mask,
                                           float(self.cfg.gaussian_scale_min), float(self.cfg.gaussian_scale_max))
        return Gaussians(means=coords, covariances=cov.reshape(*lead, 3, 3), harmonics=sh.reshape(*lead, 3, 9),
                         opacities=opacities, scales=scales.reshape(*lead, 3), rotations=rot.reshape(*lead, 4))
