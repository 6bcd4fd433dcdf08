"""CPU tests that pin the rasterizer ORACLE itself (the reference has no golden vectors for this
boundary -- SURVEY.md 8(c)): closed-form known answers, an independent float64 torch restatement,
and autograd for the hand-derived backward."""
import math

import numpy as np
import pytest
import torch

from oracle import raster_oracle as ro
from oracle.raster_dense_torch import render_dense
from util_framing import _frame
from util_raster import oracle_forward, small_scene, view_inputs

C0 = 0.28209479177387814


def test_exp_accuracy():
    L = ro.lib()
    xs = np.concatenate([np.linspace(-80, 0, 4001), -np.logspace(-8, 1.5, 500)]).astype(np.float32)
    got = np.array([L.fso_exp_public(float(x)) for x in xs], np.float64)
    ref = np.exp(xs.astype(np.float64))
    rel = np.abs(got - ref) / ref
    # alpha = opacity * exp(power) can reach the 1/255 threshold only for power >= -5.55: that is where the accuracy
    # matters (one-fma reduction: 1.7e-7 there; the |n| > 8 of the far tail add up to 2e-7 more, on alphas < 1e-3 / 255)
    assert rel[xs >= -6.0].max() < 2e-7
    assert rel.max() < 4e-7
    assert L.fso_exp_public(-100.0) == 0.0
    assert L.fso_exp_public(0.0) == 1.0


def _axis_camera(H, W, near=0.5, far=15.0):
    """Identity pose, normalised fx=fy=1 (tan(fov/2) = 0.5), framed like render_cuda does."""
    extr = torch.eye(4)[None]
    K = torch.tensor([[[1.0, 0, 0.5], [0, 1.0, 0.5], [0, 0, 1]]])
    e, scale, tx, ty, view, full = _frame(extr, K, torch.tensor([near]), torch.tensor([far]), True)
    return dict(H=H, W=W, tanfovx=float(tx[0]), tanfovy=float(ty[0]), viewmatrix=view[0], projmatrix=full[0],
                campos=e[0, :3, 3], scale=float(scale[0]))


def _iso(vi, pos, sigma, opacity, dc, bg=(0.0, 0.0, 0.0)):
    """Isotropic Gaussians at world `pos` (pre-scale), world sigma, DC-only SH."""
    s = vi["scale"]
    pos = torch.tensor(pos, dtype=torch.float32).reshape(-1, 3) * s
    n = pos.shape[0]
    sig = torch.tensor(sigma, dtype=torch.float32).reshape(n) * s
    cov = torch.zeros(n, 6)
    cov[:, 0] = cov[:, 3] = cov[:, 5] = sig ** 2
    shs = torch.zeros(n, 9, 3)
    shs[:, 0, :] = torch.tensor(dc, dtype=torch.float32).reshape(n, 3)
    d = dict(vi)
    d.update(bg=torch.tensor(bg), sh_degree=2, means3D=pos, cov3D=cov, shs=shs,
             opacities=torch.tensor(opacity, dtype=torch.float32).reshape(n))
    return d


def test_kat_single_isotropic_gaussian():
    H = W = 32
    vi = _iso(_axis_camera(H, W), [[0, 0, 2.0]], [0.08], [0.8], [[0.5, -0.2, 1.0]], bg=(0.1, 0.2, 0.3))
    st = oracle_forward(vi)
    fx = W / (2 * vi["tanfovx"])
    z = 2.0 * vi["scale"]
    var = (fx / z) ** 2 * (0.08 * vi["scale"]) ** 2 + 0.3
    ys, xs = np.mgrid[0:H, 0:W]
    d2 = (xs - 15.5) ** 2 + (ys - 15.5) ** 2
    alpha = np.minimum(0.99, 0.8 * np.exp(-0.5 * d2 / var))
    alpha[alpha < 1 / 255] = 0
    col = np.maximum(np.array([0.5, -0.2, 1.0]) * C0 + 0.5, 0)
    bg = np.array([0.1, 0.2, 0.3])
    exp_color = col[:, None, None] * alpha[None] + (1 - alpha)[None] * bg[:, None, None]
    assert st["radii"][0] == math.ceil(3 * math.sqrt(var))
    assert st["num_rendered"] == 4
    np.testing.assert_allclose(st["means2D"][0], [15.5, 15.5], atol=1e-4)
    np.testing.assert_allclose(st["color"], exp_color, atol=2e-6)
    np.testing.assert_allclose(st["depth"], z * alpha, atol=1e-5)
    np.testing.assert_allclose(st["alpha"], alpha, atol=2e-6)
    np.testing.assert_array_equal(st["n_contrib"], (alpha > 0).astype(np.int32))


def test_kat_two_gaussians_order_and_depth():
    H = W = 32
    vi = _iso(_axis_camera(H, W), [[0, 0, 3.0], [0.0, 0.0, 1.5]], [0.15, 0.06], [0.7, 0.6],
              [[1.0, 0, 0], [0, 1.0, 0]])
    st = oracle_forward(vi)
    # gaussian 1 (id 1) is nearer -> drawn first in every tile
    for t in range(4):
        a, b = st["ranges"][t]
        assert list(st["point_list"][a:b]) == [1, 0]
    s = vi["scale"]
    fx = W / (2 * vi["tanfovx"])
    ys, xs = np.mgrid[0:H, 0:W]

    def alpha_of(pos, sig, op):
        z = pos[2] * s
        var = (fx / z) ** 2 * (sig * s) ** 2 + 0.3
        cx = fx * pos[0] / pos[2] + 15.5
        cy = fx * pos[1] / pos[2] + 15.5
        a = np.minimum(0.99, op * np.exp(-0.5 * ((xs - cx) ** 2 + (ys - cy) ** 2) / var))
        a[a < 1 / 255] = 0
        return a, z

    a_near, z_near = alpha_of([0.0, 0, 1.5], 0.06, 0.6)
    a_far, z_far = alpha_of([0, 0, 3.0], 0.15, 0.7)
    c_near = np.maximum(np.array([0, 1.0, 0]) * C0 + 0.5, 0)
    c_far = np.maximum(np.array([1.0, 0, 0]) * C0 + 0.5, 0)
    exp = c_near[:, None, None] * a_near + c_far[:, None, None] * (a_far * (1 - a_near))
    np.testing.assert_allclose(st["color"], exp, atol=3e-6)
    np.testing.assert_allclose(st["depth"], z_near * a_near + z_far * a_far * (1 - a_near), atol=2e-5)
    # swapping the draw order would give a different image (order-dependent blend)
    wrong = c_far[:, None, None] * a_far + c_near[:, None, None] * (a_near * (1 - a_far))
    assert np.abs(wrong - exp).max() > 1e-2


def test_kat_near_cull_and_empty():
    H = W = 32
    cam = _axis_camera(H, W)
    # view-space z (after the 1/near rescale) <= 0.2 is culled: world z = 0.1 -> 0.2 exactly
    vi = _iso(cam, [[0, 0, 0.1], [0, 0, -1.0]], [0.05, 0.05], [0.9, 0.9], [[1, 1, 1], [1, 1, 1]], bg=(0.3, 0.1, 0.2))
    st = oracle_forward(vi)
    assert st["num_rendered"] == 0 and (st["radii"] == 0).all()
    np.testing.assert_array_equal(st["color"], np.broadcast_to(np.array([0.3, 0.1, 0.2], np.float32)[:, None, None], (3, H, W)))
    assert (st["depth"] == 0).all() and (st["n_contrib"] == 0).all() and (st["final_T"] == 1).all()


def test_transmittance_cutoff_and_alpha_cap():
    H = W = 16
    cam = _axis_camera(H, W)
    n = 12
    pos = [[0, 0, 1.0 + 0.1 * i] for i in range(n)]
    vi = _iso(cam, pos, [0.5] * n, [5.0] * n, [[1, 1, 1]] * n)  # opacity 5 -> alpha capped at 0.99
    st = oracle_forward(vi)
    # alpha = 0.99f for every Gaussian; T after the first = 1 - 0.99f = 0.00999999; the second would
    # give 0.00999999^2 = 9.99998e-5 < 1e-4 -> the pixel stops BEFORE applying it (App. A.4)
    assert (st["n_contrib"] == 1).all()
    np.testing.assert_allclose(st["final_T"], 1.0 - np.float32(0.99), rtol=1e-6)
    np.testing.assert_allclose(st["alpha"], 0.99, rtol=1e-6)


@pytest.mark.parametrize("sh_degree,precomp", [(2, False), (3, False), (0, False), (1, True)])
def test_dense_float64_restatement_matches(sh_degree, precomp):
    H, W = 40, 56
    scene, cams = small_scene(N=300, H=H, W=W, seed=11 + sh_degree, sh_degree=max(sh_degree, 0))
    vi = view_inputs(scene, cams, 1, H, W, bg=(0.2, 0.4, 0.6))
    if precomp:
        vi["colors_precomp"] = vi["shs"][:, 0, :].abs().contiguous()
        vi["shs"] = None
    st = oracle_forward(vi)
    assert st["num_rendered"] > 300
    f64 = lambda t: None if t is None else t.double()
    order = torch.from_numpy(np.lexsort((np.arange(st["N"]), st["depths"].view(np.uint32))).astype(np.int64))
    color, depth, _ = render_dense(H, W, vi["tanfovx"], vi["tanfovy"], vi["bg"], vi["viewmatrix"], vi["projmatrix"],
                                   vi["sh_degree"], vi["campos"], f64(vi["means3D"]), f64(vi["cov3D"]),
                                   f64(vi["opacities"]), shs=f64(vi.get("shs")),
                                   colors_precomp=f64(vi.get("colors_precomp")),
                                   rect=torch.from_numpy(st["rect"]), radii=torch.from_numpy(st["radii"]), order=order)
    np.testing.assert_allclose(st["color"], color.numpy(), atol=2e-5)
    np.testing.assert_allclose(st["depth"], depth.numpy(), atol=1e-4)


def test_binning_invariants():
    H, W = 72, 100  # ragged: not multiples of 16
    scene, cams = small_scene(N=2000, H=H, W=W, seed=3)
    vi = view_inputs(scene, cams, 0, H, W)
    st = oracle_forward(vi)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    assert st["ranges"].shape == (gx * gy, 2)
    assert st["ranges"][0, 0] == 0 and st["ranges"][-1, 1] == st["num_rendered"] == st["tiles_touched"].sum()
    assert (st["ranges"][1:, 0] == st["ranges"][:-1, 1]).all()
    dbits = st["depths"].view(np.uint32)
    for t in range(gx * gy):
        a, b = st["ranges"][t]
        ids = st["point_list"][a:b].astype(np.int64)
        key = dbits[ids].astype(np.int64) << 32 | ids
        assert (np.diff(key) > 0).all()
        tx, ty = t % gx, t // gx
        r = st["rect"][ids]
        assert ((r[:, 0] <= tx) & (tx < r[:, 2]) & (r[:, 1] <= ty) & (ty < r[:, 3])).all()


@pytest.mark.parametrize("sh_degree,precomp,with_depth", [(2, False, True), (3, False, False), (1, True, True)])
def test_backward_matches_autograd_of_dense(sh_degree, precomp, with_depth):
    H, W = 32, 48
    scene, cams = small_scene(N=60, H=H, W=W, seed=21 + sh_degree, sh_degree=sh_degree)
    scene["opacities"] = scene["opacities"].clamp(0.05, 0.9)
    vi = view_inputs(scene, cams, 0, H, W, bg=(0.3, 0.5, 0.1))
    if precomp:
        vi["colors_precomp"] = (vi["shs"][:, 0, :] * 0.5 + 0.5).contiguous()
        vi["shs"] = None
    st = oracle_forward(vi)
    rng = np.random.default_rng(0)
    g_color = rng.normal(size=(3, H, W)).astype(np.float32)
    g_depth = rng.normal(size=(H, W)).astype(np.float32) if with_depth else None
    got = ro.backward(st, g_color, g_depth)

    leaf = lambda t: None if t is None else t.double().clone().requires_grad_(True)
    m, c, o = leaf(vi["means3D"]), leaf(vi["cov3D"]), leaf(vi["opacities"])
    s, cp = leaf(vi.get("shs")), leaf(vi.get("colors_precomp"))
    order = torch.from_numpy(np.lexsort((np.arange(st["N"]), st["depths"].view(np.uint32))).astype(np.int64))
    color, depth, (px, py) = render_dense(H, W, vi["tanfovx"], vi["tanfovy"], vi["bg"], vi["viewmatrix"],
                                           vi["projmatrix"], vi["sh_degree"], vi["campos"], m, c, o, shs=s,
                                           colors_precomp=cp, rect=torch.from_numpy(st["rect"]),
                                           radii=torch.from_numpy(st["radii"]), order=order)
    np.testing.assert_allclose(st["color"], color.detach().numpy(), atol=2e-5)
    px.retain_grad(); py.retain_grad()
    loss = (color * torch.from_numpy(g_color).double()).sum()
    if with_depth:
        loss = loss + (depth * torch.from_numpy(g_depth).double()).sum()
    loss.backward()

    def close(a, b, name):
        b = b.numpy()
        scale = np.abs(b).max() + 1e-12
        err = np.abs(a - b).max() / scale
        assert err < 2e-4, f"{name}: rel-to-max err {err}"

    close(got["means3D"], m.grad, "means3D")
    close(got["cov3D"], c.grad, "cov3D")
    close(got["opacities"], o.grad, "opacities")
    if precomp:
        close(got["colors_precomp"], cp.grad, "colors")
    else:
        close(got["shs"], s.grad, "shs")


def test_backward_matches_finite_differences_of_dense():
    """the hand-derived backward of the C oracle against central finite differences of the
    float64 dense restatement (the discrete pieces -- tile rects, radii, depth order -- held fixed), on a scene small
    enough to perturb parameter by parameter."""
    H, W = 24, 32
    scene, cams = small_scene(N=12, H=H, W=W, seed=77, sh_degree=1)
    scene["opacities"] = scene["opacities"].clamp(0.1, 0.8)
    vi = view_inputs(scene, cams, 0, H, W, bg=(0.2, 0.1, 0.3))
    st = oracle_forward(vi)
    rng = np.random.default_rng(3)
    g_color = rng.normal(size=(3, H, W)).astype(np.float32)
    got = ro.backward(st, g_color)
    order = torch.from_numpy(np.lexsort((np.arange(st["N"]), st["depths"].view(np.uint32))).astype(np.int64))
    gc = torch.from_numpy(g_color).double()

    def loss(m, c, o, s):
        color, _, _ = render_dense(H, W, vi["tanfovx"], vi["tanfovy"], vi["bg"], vi["viewmatrix"], vi["projmatrix"],
                                   vi["sh_degree"], vi["campos"], m, c, o, shs=s, rect=torch.from_numpy(st["rect"]),
                                   radii=torch.from_numpy(st["radii"]), order=order)
        return float((color * gc).sum())

    base = [vi[k].double().clone() for k in ("means3D", "cov3D", "opacities", "shs")]
    names = ("means3D", "cov3D", "opacities", "shs")
    for which, name in enumerate(names):
        flat = base[which].reshape(-1)
        ana = np.asarray(got[name], np.float64).reshape(-1)
        idx = rng.choice(flat.numel(), size=min(12, flat.numel()), replace=False)
        fd = np.zeros(len(idx))
        for j, i in enumerate(idx):
            h = 1e-5 * max(1.0, abs(float(flat[i])))
            args_p, args_m = [t.clone() for t in base], [t.clone() for t in base]
            args_p[which].reshape(-1)[i] += h
            args_m[which].reshape(-1)[i] -= h
            fd[j] = (loss(*args_p) - loss(*args_m)) / (2 * h)
        scale = np.abs(ana).max() + 1e-12
        assert np.abs(fd - ana[idx]).max() / scale < 2e-3, (name, fd, ana[idx])
