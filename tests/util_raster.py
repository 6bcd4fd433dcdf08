"""Shared helpers for the rasterizer tests: scenes -> per-view rasterizer inputs."""
from __future__ import annotations

import numpy as np
import torch

from freesplat_amd import synthetic
from util_framing import _frame


def view_inputs(scene: dict, cams: dict, i: int, H: int, W: int, bg=(0.0, 0.0, 0.0)) -> dict:
    """Inputs of ONE rasterizer call, framed exactly as render_cuda frames them (CPU tensors)."""
    extr, scale, tan_x, tan_y, view, full = _frame(cams["extrinsics"], cams["intrinsics"], cams["near"],
                                                   cams["far"], True)
    s = scale[i]
    means = scene["means"] * s
    cov = scene["covariances"] * (s * s)
    r, c = torch.triu_indices(3, 3)
    shs = scene["harmonics"].transpose(-1, -2).contiguous()
    return dict(H=H, W=W, tanfovx=float(tan_x[i]), tanfovy=float(tan_y[i]),
                bg=torch.tensor(bg, dtype=torch.float32), viewmatrix=view[i].contiguous(),
                projmatrix=full[i].contiguous(), campos=extr[i, :3, 3].contiguous(),
                sh_degree=int(round(shs.shape[1] ** 0.5)) - 1,
                means3D=means.contiguous(), cov3D=cov[:, r, c].contiguous(), shs=shs,
                opacities=scene["opacities"].contiguous())


def small_scene(N=600, H=64, W=80, seed=7, n_views=2, sh_degree=2):
    scene = synthetic.make_scene(N, n_context=2, seed=seed, sh_degree=sh_degree, ctx_hw=(H, W))
    cams = synthetic.target_cameras(n_views, seed=seed)
    return scene, cams


def oracle_forward(vi: dict, **kw):
    from oracle import raster_oracle as ro
    n = lambda t: t.detach().cpu().numpy()
    return ro.forward(vi["H"], vi["W"], vi["tanfovx"], vi["tanfovy"], n(vi["bg"]), n(vi["viewmatrix"]),
                      n(vi["projmatrix"]), vi["sh_degree"], n(vi["campos"]), n(vi["means3D"]),
                      n(vi["cov3D"]), n(vi["opacities"]),
                      shs=None if vi.get("shs") is None else n(vi["shs"]),
                      colors_precomp=None if vi.get("colors_precomp") is None else n(vi["colors_precomp"]),
                      **kw)


def hip_forward(vi: dict, device, requires_grad=False):
    """Run the product rasterizer on `device`; returns (outputs tuple, leaf tensors dict)."""
    from freesplat_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    d = lambda t: None if t is None else t.to(device)
    leaves = {}
    for k in ("means3D", "cov3D", "shs", "colors_precomp", "opacities"):
        t = vi.get(k)
        if t is not None:
            t = t.to(device).clone().requires_grad_(requires_grad)
        leaves[k] = t
    s = GaussianRasterizationSettings(vi["H"], vi["W"], vi["tanfovx"], vi["tanfovy"], d(vi["bg"]), 1.0,
                                      d(vi["viewmatrix"]), d(vi["projmatrix"]), vi["sh_degree"],
                                      d(vi["campos"]), False, False)
    means2D = torch.zeros(leaves["means3D"].shape[0], 3, device=device, requires_grad=requires_grad)
    leaves["means2D"] = means2D
    out = GaussianRasterizer(s)(means3D=leaves["means3D"], means2D=means2D, shs=leaves["shs"],
                                colors_precomp=leaves["colors_precomp"], opacities=leaves["opacities"][:, None]
                                if leaves["opacities"].dim() == 1 else leaves["opacities"],
                                cov3D_precomp=leaves["cov3D"])
    return out, leaves
