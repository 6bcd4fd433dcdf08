"""Independent float64, autograd-differentiable, NON-tiled restatement of the rasterizer math.

TEST INFRASTRUCTURE ONLY (see oracle/raster_oracle.c header).  Used to cross-check the
hand-derived backward of the C oracle (and through it the HIP kernels) on tiny scenes
(<= ~100 Gaussians, <= 48x48 px): every pixel evaluates every Gaussian, O(N*P) memory.

The discrete parts (tile rectangles, draw order) are taken from the C oracle's forward so
that both evaluate exactly the same (gaussian, pixel) pairs; everything differentiable is
re-derived here from SURVEY.md Appendix A (3DGS paper, EWA splatting) using torch ops only.
Gradient conventions of the original that are NOT the mathematical derivative are emulated
with detach() tricks and called out inline.
"""
from __future__ import annotations

import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
      -0.4570457994644658, 1.445305721320277, -0.5900435899266435)


def _sh_color(deg, dirs, shs):
    x, y, z = dirs.unbind(-1)
    x, y, z = x[:, None], y[:, None], z[:, None]
    res = C0 * shs[:, 0]
    if deg > 0:
        res = res - C1 * y * shs[:, 1] + C1 * z * shs[:, 2] - C1 * x * shs[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + C2[0] * xy * shs[:, 4] + C2[1] * yz * shs[:, 5] + C2[2] * (2 * zz - xx - yy) * shs[:, 6]
               + C2[3] * xz * shs[:, 7] + C2[4] * (xx - yy) * shs[:, 8])
    if deg > 2:
        res = (res + C3[0] * y * (3 * xx - yy) * shs[:, 9] + C3[1] * xy * z * shs[:, 10]
               + C3[2] * y * (4 * zz - xx - yy) * shs[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * shs[:, 12]
               + C3[4] * x * (4 * zz - xx - yy) * shs[:, 13] + C3[5] * z * (xx - yy) * shs[:, 14]
               + C3[6] * x * (xx - 3 * yy) * shs[:, 15])
    return res + 0.5


def render_dense(H, W, tanfovx, tanfovy, bg, viewmatrix, projmatrix, sh_degree, campos,
                 means3D, cov6, opacities, shs=None, colors_precomp=None, *, rect, radii, order):
    """All tensor args float64.  rect [N,4] int (tile rect), radii [N] int, order: LongTensor of
    gaussian ids sorted by (depth bits, id) -- all from the C oracle.  Returns color[3,H,W], depth[H,W]."""
    dt = torch.float64
    N = means3D.shape[0]
    V = viewmatrix.to(dt)
    M = projmatrix.to(dt)
    ones = torch.ones(N, 1, dtype=dt)
    ph = torch.cat([means3D, ones], -1)
    pv = ph @ V  # row-vector convention: torch viewmatrix is world->cam transposed
    phom = ph @ M
    pw = 1.0 / (phom[:, 3] + 1e-7)
    ndc = phom[:, :2] * pw[:, None]
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    tz = pv[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    txtz, tytz = pv[:, 0] / tz, pv[:, 1] / tz
    cx_ = (txtz < -limx) | (txtz > limx)
    cy_ = (tytz < -limy) | (tytz > limy)
    # original: clamped coordinate gets zero gradient and its tz dependence is ignored
    tx = torch.where(cx_, (txtz.clamp(-limx, limx) * tz).detach(), pv[:, 0])
    ty = torch.where(cy_, (tytz.clamp(-limy, limy) * tz).detach(), pv[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([torch.stack([fx / tz, zero, -fx * tx / (tz * tz)], -1),
                     torch.stack([zero, fy / tz, -fy * ty / (tz * tz)], -1)], -2)  # [N,2,3]
    R = V[:3, :3].T  # world->cam rotation (standard, column-vector)
    Sig = torch.zeros(N, 3, 3, dtype=dt)
    idx = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
    for k, (i, j) in enumerate(idx):
        Sig[:, i, j] = cov6[:, k]
        Sig[:, j, i] = cov6[:, k]
    Mm = J @ R
    cov2 = Mm @ Sig @ Mm.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c = cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    conA, conB, conC = c / det, -b / det, a / det
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    if shs is not None:
        d = means3D - campos.to(dt)[None]
        d = d / d.norm(dim=-1, keepdim=True)
        col = _sh_color(sh_degree, d, shs)
        col = torch.clamp_min(col, 0.0)  # zero grad where clamped, as the original
    else:
        col = colors_precomp

    order = order.long()
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    pxs, pys = xs.reshape(-1), ys.reshape(-1)  # [P]
    tile_x = (pxs // 16).long()
    tile_y = (pys // 16).long()
    o = order
    rx0, ry0, rx1, ry1 = [rect[o, k].long()[:, None] for k in range(4)]
    member = (tile_x[None] >= rx0) & (tile_x[None] < rx1) & (tile_y[None] >= ry0) & (tile_y[None] < ry1)
    member = member & (radii[o] > 0)[:, None]
    dx = px[o][:, None] - pxs[None]
    dy = py[o][:, None] - pys[None]
    power = -0.5 * (conA[o][:, None] * dx * dx + conC[o][:, None] * dy * dy) - conB[o][:, None] * dx * dy
    G = torch.exp(power)
    raw = opacities[o][:, None] * G
    # original: alpha = min(0.99, o*G) but the backward ignores the clamp
    alpha = raw + (torch.clamp_max(raw, 0.99) - raw).detach()
    active = member & (power <= 0) & (alpha >= 1.0 / 255.0)
    a_eff = torch.where(active, alpha, torch.zeros_like(alpha))
    one_m = 1.0 - a_eff
    T_incl = torch.cumprod(one_m, 0)
    stop = active & (T_incl < 1e-4)
    done = torch.cummax(stop.to(torch.int8), 0).values.bool()
    a_eff = torch.where(done, torch.zeros_like(a_eff), a_eff)
    one_m = 1.0 - a_eff
    T_incl = torch.cumprod(one_m, 0)
    T_excl = torch.cat([torch.ones(1, T_incl.shape[1], dtype=dt), T_incl[:-1]], 0)
    w = a_eff * T_excl  # [N,P]
    color = torch.einsum("np,nc->cp", w, col[o]) + T_incl[-1][None] * bg.to(dt)[:, None]
    depth = (w * tz[o][:, None]).sum(0)
    return color.reshape(3, H, W), depth.reshape(H, W), (px, py)
