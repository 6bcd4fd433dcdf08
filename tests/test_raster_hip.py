"""GPU parity tests of the HIP rasterizer (through the C ABI) against the CPU oracle.

Bar (BASELINE.json north_star): <= 1e-4 abs per pixel with identical tile/depth ordering.
What is asserted here is stronger for the forward: bit-exact images, radii, tile ranges and
depth-sorted id lists (the oracle and the kernels share one arithmetic contract).  The backward
accumulates with float atomics, so its bar is a tolerance: 2e-4 of the gradient's max-abs.
"""
import numpy as np
import pytest
import torch

from freesplat_amd import synthetic
from util_raster import hip_forward, oracle_forward, small_scene, view_inputs

pytestmark = pytest.mark.gpu

ATOL_PIXEL = 1e-4  # the north_star tolerance, fp32


def _check_forward(vi, device, exact=True):
    from freesplat_amd.rasterizer import debug_state
    st = oracle_forward(vi)
    (color, radii, depth, alpha), leaves = hip_forward(vi, device)
    rs = color.grad_fn.rs if color.grad_fn is not None else None
    c, d, a = color.cpu().numpy(), depth.cpu().numpy(), alpha.cpu().numpy()
    assert np.abs(c - st["color"]).max() <= ATOL_PIXEL
    assert np.abs(d - st["depth"]).max() <= ATOL_PIXEL * max(1.0, np.abs(st["depth"]).max())
    np.testing.assert_array_equal(radii.cpu().numpy(), st["radii"])
    if exact:
        np.testing.assert_array_equal(c, st["color"])
        np.testing.assert_array_equal(d, st["depth"])
        np.testing.assert_array_equal(a, st["alpha"])
    return st, (color, radii, depth, alpha), leaves


def _internal_state(vi, device):
    """Forward with grad enabled so the autograd node (and its RasterState) is reachable."""
    from freesplat_amd.rasterizer import debug_state
    out, leaves = hip_forward(vi, device, requires_grad=True)
    return debug_state(out[0].grad_fn.rs), out, leaves


def _set_cull(monkeypatch, on):
    from freesplat_amd import rasterizer as R
    monkeypatch.setattr(R, "TILE_CULL", on)


def _check_lists(dbg, st, W, H, cull, sample=None):
    """cull off: tile ranges and id lists equal the oracle's (= reference semantics).
    cull on : every tile list is an order-preserving subsequence of the oracle's, and every dropped
    (gaussian, tile) instance has alpha < 1/255 on all 256 pixels of its tile (float64 check)."""
    if not cull:
        assert dbg["num_rendered"] == st["num_rendered"]
        np.testing.assert_array_equal(dbg["offsets"][:-1], st["ranges"][:, 0])
        np.testing.assert_array_equal(dbg["offsets"][1:], st["ranges"][:, 1])
        np.testing.assert_array_equal(dbg["point_list"], st["point_list"])
        np.testing.assert_array_equal(dbg["n_contrib"], st["n_contrib"])
        return 1.0
    T = st["ranges"].shape[0]
    g_tile = np.repeat(np.arange(T, dtype=np.int64), np.diff(dbg["offsets"].astype(np.int64)))
    o_tile = np.repeat(np.arange(T, dtype=np.int64), (st["ranges"][:, 1] - st["ranges"][:, 0]).astype(np.int64))
    g_id, o_id = dbg["point_list"].astype(np.int64), st["point_list"].astype(np.int64)
    g_comp, o_comp = (g_tile << 32) | g_id, (o_tile << 32) | o_id
    kept = np.isin(o_comp, g_comp)
    assert np.isin(g_comp, o_comp).all() and kept.sum() == len(g_comp)
    np.testing.assert_array_equal(o_comp[kept], g_comp)  # same relative order
    rm = np.nonzero(~kept)[0]
    if sample is not None and len(rm) > sample:
        rm = np.random.default_rng(0).choice(rm, sample, replace=False)
    gx = (W + 15) // 16
    ids, tiles = o_id[rm], o_tile[rm]
    ys, xs = np.mgrid[0:16, 0:16]
    pxs = (tiles % gx)[:, None] * 16 + xs.reshape(1, -1)
    pys = (tiles // gx)[:, None] * 16 + ys.reshape(1, -1)
    co = st["conic_opacity"].astype(np.float64)[ids]
    dx = st["means2D"].astype(np.float64)[ids, 0:1] - pxs
    dy = st["means2D"].astype(np.float64)[ids, 1:2] - pys
    power = -0.5 * (co[:, 0:1] * dx * dx + co[:, 2:3] * dy * dy) - co[:, 1:2] * dx * dy
    alpha = co[:, 3:4] * np.exp(np.minimum(power, 0.0))
    inside = (pxs < W) & (pys < H)
    assert (alpha[inside & (power <= 0)] < 1.0 / 255.0).all()
    return len(g_comp) / max(len(o_comp), 1)


@pytest.mark.parametrize("cull", [False, True])
@pytest.mark.parametrize("H,W,N,seed", [(64, 80, 600, 7), (72, 100, 3000, 3), (256, 256, 20000, 5), (16, 16, 50, 1)])
def test_forward_bit_exact_and_ordering(hip_device, monkeypatch, H, W, N, seed, cull):
    _set_cull(monkeypatch, cull)
    scene, cams = small_scene(N=N, H=H, W=W, seed=seed)
    vi = view_inputs(scene, cams, 1, H, W, bg=(0.1, 0.2, 0.3))
    st, _, _ = _check_forward(vi, hip_device)
    dbg, _, _ = _internal_state(vi, hip_device)
    _check_lists(dbg, st, W, H, cull)  # identical tile/depth ordering
    np.testing.assert_array_equal(dbg["final_T"], st["final_T"])
    np.testing.assert_array_equal(dbg["rect"].astype(np.int32), st["rect"])
    np.testing.assert_array_equal(dbg["rec"][:, 0:2], st["means2D"])
    np.testing.assert_array_equal(dbg["rec"][:, 8:11], st["rgb"])
    np.testing.assert_array_equal(dbg["rec"][:, 7], st["depths"])


@pytest.mark.parametrize("sh_degree", [0, 1, 2, 3])
def test_forward_sh_degrees(hip_device, sh_degree):
    scene, cams = small_scene(N=800, H=48, W=64, seed=30 + sh_degree, sh_degree=sh_degree)
    vi = view_inputs(scene, cams, 0, 48, 64)
    _check_forward(vi, hip_device)


def test_forward_colors_precomp(hip_device):
    scene, cams = small_scene(N=800, H=48, W=64, seed=9)
    vi = view_inputs(scene, cams, 0, 48, 64, bg=(1.0, 1.0, 1.0))
    vi["colors_precomp"] = (vi["shs"][:, 0, :] * 0.5 + 0.5).contiguous()
    vi["shs"] = None
    _check_forward(vi, hip_device)


def test_empty_and_all_culled(hip_device):
    scene, cams = small_scene(N=100, H=32, W=32, seed=2)
    vi = view_inputs(scene, cams, 0, 32, 32, bg=(0.3, 0.1, 0.2))
    behind = dict(vi)
    behind["means3D"] = vi["means3D"] * torch.tensor([1.0, 1.0, -1.0])
    st, (color, radii, depth, alpha), _ = _check_forward(behind, hip_device)
    assert st["num_rendered"] == 0 and (radii == 0).all()
    assert torch.equal(color.cpu(), torch.tensor([0.3, 0.1, 0.2])[:, None, None].expand(3, 32, 32))
    empty = dict(vi)
    for k, shape in (("means3D", (0, 3)), ("cov3D", (0, 6)), ("shs", (0, 9, 3)), ("opacities", (0,))):
        empty[k] = torch.zeros(shape)
    (color, radii, depth, alpha), _ = hip_forward(empty, hip_device)
    assert radii.numel() == 0 and (depth == 0).all() and (alpha == 0).all()
    assert torch.equal(color.cpu(), torch.tensor([0.3, 0.1, 0.2])[:, None, None].expand(3, 32, 32))


def test_long_tile_lists_global_sort_path(hip_device, monkeypatch):
    """> 4096 instances in one tile: exercises the in-HBM sort fallback and many render rounds."""
    _set_cull(monkeypatch, False)
    H = W = 32
    scene, cams = small_scene(N=6000, H=H, W=W, seed=4)
    scene["covariances"] = scene["covariances"] * 400.0  # every Gaussian covers the whole image
    scene["opacities"] = scene["opacities"] * 0.02        # keep transmittance alive through the list
    vi = view_inputs(scene, cams, 0, H, W)
    st, _, _ = _check_forward(vi, hip_device)
    assert (st["ranges"][:, 1] - st["ranges"][:, 0]).max() > 4096
    dbg, _, _ = _internal_state(vi, hip_device)
    assert np.diff(dbg["offsets"].astype(np.int64)).max() > 4096
    _check_lists(dbg, st, W, H, False)


@pytest.mark.parametrize("N", [600, 1150, 1400, 1900, 2300, 2900, 3700])
def test_tile_sort_size_classes(hip_device, monkeypatch, N):
    """Every size class of the per-tile sort (single network at 2/4/8/16 keys per thread and the two-run forms
    for lists a little above a power of two): N Gaussians that all cover the single 16x16 tile."""
    _set_cull(monkeypatch, False)
    H = W = 16
    scene, cams = small_scene(N=N, H=H, W=W, seed=40 + N)
    scene["covariances"] = scene["covariances"] * 400.0
    scene["opacities"] = scene["opacities"] * 0.01
    vi = view_inputs(scene, cams, 0, H, W)
    st, _, _ = _check_forward(vi, hip_device)
    n = int(st["ranges"][0, 1] - st["ranges"][0, 0])
    lo, hi = {600: (513, 768), 1150: (1025, 1280), 1400: (1281, 1536), 1900: (1537, 2048), 2300: (2049, 2560),
              2900: (2561, 3072), 3700: (3073, 4096)}[N]
    assert lo <= n <= hi, n
    dbg, _, _ = _internal_state(vi, hip_device)
    _check_lists(dbg, st, W, H, False)


def test_depth_ties_break_by_index(hip_device, monkeypatch):
    _set_cull(monkeypatch, False)
    H = W = 32
    scene, cams = small_scene(N=64, H=H, W=W, seed=6)
    # duplicate every Gaussian: identical depth keys, order must follow the index
    for k in ("means", "covariances", "harmonics", "opacities"):
        scene[k] = torch.cat([scene[k], scene[k]])
    vi = view_inputs(scene, cams, 0, H, W)
    st, _, _ = _check_forward(vi, hip_device)
    dbg, _, _ = _internal_state(vi, hip_device)
    np.testing.assert_array_equal(dbg["point_list"], st["point_list"])


@pytest.mark.parametrize("copies,N", [(30, 40), (3, 700)])
def test_tile_sort_heavy_depth_ties(hip_device, monkeypatch, copies, N):
    """Many Gaussians with bit-identical depth in one tile: the tile sort's depth buckets overflow (30 copies -> the
    bitonic fallback) or hold multi-way ties (3 copies, resolved by the in-bucket 64-bit compare); either way the
    list order is (depth, index) exactly as the oracle's stable sort gives it."""
    _set_cull(monkeypatch, False)
    H = W = 32
    scene, cams = small_scene(N=N, H=H, W=W, seed=16)
    for k in ("means", "covariances", "harmonics", "opacities"):
        scene[k] = torch.cat([scene[k]] * copies)
    vi = view_inputs(scene, cams, 0, H, W)
    st, _, _ = _check_forward(vi, hip_device)
    dbg, _, _ = _internal_state(vi, hip_device)
    np.testing.assert_array_equal(dbg["point_list"], st["point_list"])
    assert (np.diff(dbg["offsets"]) > 24).any()


@pytest.mark.parametrize("copies,N", [(3, 2500), (2600, 2), (40, 150)])
def test_long_list_partitioned_sort_with_depth_ties(hip_device, monkeypatch, copies, N):
    """Tile lists beyond the LDS sort (the two-level distribution sort, round 5) with bit-identical depths: triples spread over
    7 500 entries (ties inside the groups' buckets), ONE depth value 2 600 times per Gaussian (a single fine bin longer than a
    group: the global-memory network fallback), and 40-fold ties at 6 000 entries (groups whose LDS sort falls back to its
    register network).  List order = (depth, index), images bit-exact."""
    _set_cull(monkeypatch, False)
    H = W = 32
    scene, cams = small_scene(N=N, H=H, W=W, seed=21)
    scene["covariances"] = scene["covariances"] * 400.0
    scene["opacities"] = scene["opacities"] * 0.01
    for k in ("means", "covariances", "harmonics", "opacities"):
        scene[k] = torch.cat([scene[k]] * copies)
    vi = view_inputs(scene, cams, 0, H, W)
    st, _, _ = _check_forward(vi, hip_device)
    assert (st["ranges"][:, 1] - st["ranges"][:, 0]).max() > 2048
    dbg, _, _ = _internal_state(vi, hip_device)
    np.testing.assert_array_equal(dbg["point_list"], st["point_list"])
    np.testing.assert_array_equal(dbg["n_contrib"], st["n_contrib"])


def test_long_list_more_groups_than_the_partitioned_sort_holds(hip_device, monkeypatch):
    """140 000 entries in ONE tile: more than 64 groups of <= 2048 keys, so sort_tile_partitioned declines and the tile is sorted
    by the global-memory network; same list as the oracle's stable sort."""
    _set_cull(monkeypatch, False)
    H = W = 16
    scene, cams = small_scene(N=140_000, H=H, W=W, seed=22)
    scene["covariances"] = scene["covariances"] * 900.0
    scene["opacities"] = scene["opacities"] * 0.01
    vi = view_inputs(scene, cams, 0, H, W)
    st, _, _ = _check_forward(vi, hip_device)
    assert int(st["ranges"][0, 1] - st["ranges"][0, 0]) > 64 * 2048
    dbg, _, _ = _internal_state(vi, hip_device)
    np.testing.assert_array_equal(dbg["point_list"], st["point_list"])


def test_capacity_overflow_retry(hip_device, monkeypatch):
    from freesplat_amd import rasterizer as R
    scene, cams = small_scene(N=3000, H=64, W=64, seed=8)
    vi = view_inputs(scene, cams, 0, 64, 64)
    monkeypatch.setattr(R, "default_capacity", lambda N, st, H=0, W=0: 100)
    st, _, _ = _check_forward(vi, hip_device)
    assert st["num_rendered"] > 100


def test_tile_key_area_overflow_retry(hip_device, monkeypatch):
    """Every tile owns a fixed key area (tile_capacity(cap, T) slots, >= 4x the mean list length): a scene that piles more
    instances onto ONE tile than that -- while the instance total still fits the capacity -- must be reported
    (counters[1] = the largest tile list) and retried with a capacity whose key areas hold it; same image as the oracle."""
    from freesplat_amd import rasterizer as R
    H = W = 128                                                   # 64 tiles
    scene, cams = small_scene(N=2, H=H, W=W, seed=12)
    for k in ("means", "covariances", "harmonics", "opacities"):
        scene[k] = torch.cat([scene[k]] * 1500)                   # 3000 Gaussians on the same few tiles
    vi = view_inputs(scene, cams, 0, H, W)
    st = oracle_forward(vi)                                       # (the oracle's lists are the unculled ones)
    counts = (st["ranges"][:, 1] - st["ranges"][:, 0]).astype(np.int64)
    cap = int(counts.sum()) + 64
    tile_cap = 2048
    while tile_cap < (4 * cap + 63) // 64:
        tile_cap *= 2
    assert counts.max() > tile_cap                                # the total fits `cap`, the big tiles' key areas do not
    monkeypatch.setattr(R, "TILE_CULL", False)
    monkeypatch.setattr(R, "default_capacity", lambda N, s, H=0, W=0: max(cap, s.retry_cap))
    R._state(hip_device).retry_cap = 0
    (color, _, _, _), _ = hip_forward(vi, hip_device)
    assert R._state(hip_device).retry_cap >= (int(counts.max()) * 64 + 3) // 4       # the retry path was taken
    np.testing.assert_array_equal(color.cpu().numpy(), st["color"])
    R._state(hip_device).retry_cap = 0


def test_render_views_overflow_now_and_deferred(hip_device, monkeypatch):
    """Capacity overflow on the batched path: check="now" re-renders the overflowed views (forward and backward then
    go view by view) and gives the same images and gradients; check="deferred" reports it at check_deferred()."""
    from freesplat_amd import _lib, rasterizer as R
    from freesplat_amd.decoder import check_deferred, render_views
    H, W, v = 48, 64, 3
    scene, cams = small_scene(N=1500, H=H, W=W, seed=21, n_views=v)
    dev = hip_device
    g = {k: scene[k].to(dev).requires_grad_(True) for k in ("means", "covariances", "harmonics", "opacities")}
    cam = {k: t.to(dev) for k, t in cams.items()}
    bg = torch.zeros(v, 3, device=dev)
    args = (cam["extrinsics"], cam["intrinsics"], cam["near"], cam["far"], (H, W), bg, g["means"], g["covariances"],
            g["harmonics"], g["opacities"])
    c_ref, d_ref = render_views(*args)
    w = torch.randn_like(c_ref)
    (c_ref * w).sum().backward()
    ref = {k: t.grad.clone() for k, t in g.items()}
    for t in g.values():
        t.grad = None
    monkeypatch.setattr(R, "default_capacity", lambda N, st, H=0, W=0: 64)
    c2, d2 = render_views(*args)                       # overflows, re-rendered view by view
    assert torch.equal(c2, c_ref) and torch.equal(d2, d_ref)
    (c2 * w).sum().backward()
    for k in g:
        assert (g[k].grad - ref[k]).abs().max() <= 2e-4 * (ref[k].abs().max() + 1e-20), k
    with torch.no_grad():
        render_views(*args, check="deferred")
        with pytest.raises(_lib.FreeSplatHipError):
            check_deferred()
        monkeypatch.undo()
        c3, _ = render_views(*args, check="deferred")  # capacity history now covers the scene
        check_deferred()
        assert torch.equal(c3, c_ref)


def test_deferred_overflow_backward_is_safe(hip_device, monkeypatch):
    """check="deferred" with gradients: the natural order is forward, loss.backward(), THEN check_deferred().  When a
    view overflowed its instance capacity the forward left no image and no valid tile lists; the backward must not
    walk them (out-of-bounds ids) -- it yields zero gradients for that view and the check still raises."""
    from freesplat_amd import _lib, rasterizer as R
    from freesplat_amd.decoder import check_deferred, render_views
    H, W, v = 48, 64, 2
    scene, cams = small_scene(N=1500, H=H, W=W, seed=22, n_views=v)
    dev = hip_device
    g = {k: scene[k].to(dev).requires_grad_(True) for k in ("means", "covariances", "harmonics", "opacities")}
    cam = {k: t.to(dev) for k, t in cams.items()}
    monkeypatch.setattr(R, "default_capacity", lambda N, st, H=0, W=0: 64)
    color, depth = render_views(cam["extrinsics"], cam["intrinsics"], cam["near"], cam["far"], (H, W),
                                torch.zeros(v, 3, device=dev), g["means"], g["covariances"], g["harmonics"],
                                g["opacities"], check="deferred")
    (torch.ones_like(color) * color).sum().backward()        # before the check: must be harmless
    torch.cuda.synchronize()
    for k, t in g.items():
        assert t.grad is not None and not t.grad.any(), k     # zero, finite, no garbage
    with pytest.raises(_lib.FreeSplatHipError):
        check_deferred()


@pytest.mark.parametrize("mode", ["depth", "disparity", "relative_disparity", "log"])
def test_render_depth_cuda(hip_device, mode):
    """render_depth_cuda / DecoderSplattingCUDA.render_depth (cuda_splatting.py:238-280, decoder_splatting_cuda.py:77-100):
    the per-Gaussian camera-space depth (or its transform) rendered as a pre-computed colour == the oracle rasterizing
    the same colours."""
    from freesplat_amd.decoder import DecoderSplattingCUDA, Gaussians, depth_to_relative_disparity, render_depth_cuda
    H, W, v = 48, 64, 2
    scene, cams = small_scene(N=900, H=H, W=W, seed=33, n_views=v)
    dev = hip_device
    cam = {k: t.to(dev) for k, t in cams.items()}
    rep = lambda t: t.to(dev)[None].expand(v, *t.shape).contiguous()
    out = render_depth_cuda(cam["extrinsics"], cam["intrinsics"], cam["near"], cam["far"], (H, W), rep(scene["means"]),
                            rep(scene["covariances"]), rep(scene["opacities"]), mode=mode)
    assert out.shape == (v, H, W)
    for i in range(v):
        vi = view_inputs(scene, cams, i, H, W)
        homog = torch.cat([scene["means"], torch.ones(scene["means"].shape[0], 1)], -1)
        z = (torch.linalg.inv(cams["extrinsics"][i]) @ homog.T)[2]
        n, f = cams["near"][i], cams["far"][i]
        fake = {"depth": z, "disparity": 1 / z, "relative_disparity": depth_to_relative_disparity(z, n, f),
                "log": z.minimum(n).maximum(f).log()}[mode]
        vi["colors_precomp"] = fake[:, None].expand(-1, 3).contiguous()
        vi["shs"] = None
        st = oracle_forward(vi)
        ref = st["color"].mean(axis=0)
        assert np.abs(out[i].detach().cpu().numpy() - ref).max() <= 1e-4 * max(1.0, float(np.abs(ref).max())), (mode, i)
    dec = DecoderSplattingCUDA(None, None).to(dev)
    g = Gaussians(scene["means"].to(dev)[None], scene["covariances"].to(dev)[None], scene["harmonics"].to(dev)[None],
                  scene["opacities"].to(dev)[None])
    d2 = dec.render_depth(g, cam["extrinsics"][None], cam["intrinsics"][None], cam["near"][None], cam["far"][None], (H, W), mode)
    assert d2.shape == (1, v, H, W) and torch.equal(d2[0].detach(), out.detach())


def test_empty_inputs_forward_backward(hip_device):
    """N == 0 (torch hands out NULL data pointers) and v == 0: background image, empty gradients, no error."""
    from freesplat_amd.decoder import render_views
    dev = hip_device
    H, W = 32, 48
    _, cams = small_scene(N=10, H=H, W=W, seed=3, n_views=2)
    cam = {k: t.to(dev) for k, t in cams.items()}
    e = lambda *s: torch.zeros(*s, device=dev, requires_grad=True)
    means, cov, sh, op = e(0, 3), e(0, 3, 3), e(0, 3, 9), e(0)
    bg = torch.tensor([[0.25, 0.5, 0.75]] * 2, device=dev)
    color, depth = render_views(cam["extrinsics"], cam["intrinsics"], cam["near"], cam["far"], (H, W), bg, means, cov, sh, op)
    assert torch.equal(color, bg[:, :, None, None].expand(2, 3, H, W)) and not depth.any()
    color.sum().backward()
    assert means.grad.shape == (0, 3) and sh.grad.shape == (0, 3, 9)
    vi = view_inputs(*small_scene(N=10, H=H, W=W, seed=3), 0, H, W)
    for k in ("means3D", "cov3D", "shs", "opacities"):
        vi[k] = vi[k][:0]
    (c, radii, d, a), leaves = hip_forward(vi, dev, requires_grad=True)
    c.sum().backward()
    assert leaves["means3D"].grad.shape == (0, 3) and radii.shape == (0,)


@pytest.mark.parametrize("precomp,with_depth,H,W,N", [(False, True, 48, 64, 500), (True, False, 40, 40, 300),
                                                      (False, False, 128, 160, 8000)])
def test_backward_matches_oracle(hip_device, precomp, with_depth, H, W, N):
    from oracle import raster_oracle as ro
    scene, cams = small_scene(N=N, H=H, W=W, seed=13)
    vi = view_inputs(scene, cams, 1, H, W, bg=(0.3, 0.5, 0.1))
    if precomp:
        vi["colors_precomp"] = (vi["shs"][:, 0, :] * 0.5 + 0.5).contiguous()
        vi["shs"] = None
    st = oracle_forward(vi)
    rng = np.random.default_rng(1)
    g_color = rng.normal(size=(3, H, W)).astype(np.float32)
    g_depth = rng.normal(size=(H, W)).astype(np.float32) if with_depth else None
    ref = ro.backward(st, g_color, g_depth)
    (color, radii, depth, alpha), leaves = hip_forward(vi, hip_device, requires_grad=True)
    loss = (color * torch.from_numpy(g_color).to(hip_device)).sum()
    if with_depth:
        loss = loss + (depth * torch.from_numpy(g_depth).to(hip_device)).sum()
    loss.backward()

    def close(t, b, name):
        a = t.grad.cpu().numpy().reshape(b.shape)
        scale = np.abs(b).max() + 1e-20
        err = np.abs(a - b).max() / scale
        assert err < 2e-4, f"{name}: error {err} of max-abs"

    close(leaves["means3D"], ref["means3D"], "means3D")
    close(leaves["cov3D"], ref["cov3D"], "cov3D")
    close(leaves["opacities"], ref["opacities"], "opacities")
    close(leaves["colors_precomp" if precomp else "shs"], ref["colors_precomp" if precomp else "shs"], "colour")
    m2 = leaves["means2D"].grad.cpu().numpy()
    assert np.abs(m2[:, :2] - ref["means2D"]).max() <= 2e-4 * (np.abs(ref["means2D"]).max() + 1e-20)
    assert (m2[:, 2] == 0).all()


def test_render_views_equals_render_cuda_and_reference_framing(hip_device):
    from freesplat_amd.decoder import DecoderSplattingCUDA, Gaussians, render_cuda, render_views
    H, W, v = 48, 64, 3
    scene, cams = small_scene(N=900, H=H, W=W, seed=17, n_views=v)
    dev = hip_device
    g = {k: scene[k].to(dev).requires_grad_(True) for k in ("means", "covariances", "harmonics", "opacities")}
    cam = {k: t.to(dev) for k, t in cams.items()}
    bg = torch.tensor([0.2, 0.3, 0.4], device=dev)[None].expand(v, 3)
    c1, d1 = render_views(cam["extrinsics"], cam["intrinsics"], cam["near"], cam["far"], (H, W), bg,
                          g["means"], g["covariances"], g["harmonics"], g["opacities"])
    w = torch.randn_like(c1)
    (c1 * w).sum().backward()
    grads1 = {k: t.grad.clone() for k, t in g.items()}
    for t in g.values():
        t.grad = None
    rep = lambda t: t[None].expand(v, *t.shape)
    c2, d2 = render_cuda(cam["extrinsics"], cam["intrinsics"], cam["near"], cam["far"], (H, W), bg,
                         rep(g["means"]), rep(g["covariances"]), rep(g["harmonics"]), rep(g["opacities"]))
    # render_views frames with fs_frame_views, render_cuda with the reference's torch ops: ulp-level different matrices
    assert (c1 - c2).abs().max() <= ATOL_PIXEL and (d1 - d2).abs().max() <= 1e-3 * d2.abs().max()
    # ... but given the SAME matrices the batched path and the rasterizer API agree bit for bit
    from freesplat_amd.decoder import frame_views
    from freesplat_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    campos, scale, tanfov, view, full = frame_views(cam["extrinsics"], cam["intrinsics"], cam["near"], cam["far"])
    r_, c_ = torch.triu_indices(3, 3)
    with torch.no_grad():
        for i in range(v):
            s = GaussianRasterizationSettings(H, W, float(tanfov[i, 0]), float(tanfov[i, 1]), bg[i], 1.0, view[i], full[i], 2,
                                              campos[i], False, False)
            ci, _, di, _ = GaussianRasterizer(s)(means3D=g["means"] * scale[i], means2D=None,
                                                 shs=g["harmonics"].transpose(-1, -2).contiguous(),
                                                 opacities=g["opacities"][:, None],
                                                 cov3D_precomp=(g["covariances"] * scale[i] ** 2)[:, r_, c_])
            assert torch.equal(ci, c1[i]) and torch.equal(di, d1[i, 0])
    (c2 * w).sum().backward()
    for k in g:
        s = grads1[k].abs().max() + 1e-20
        assert (g[k].grad - grads1[k]).abs().max() / s < 2e-4, k
    # each view against the oracle, framed by the same host code on CPU
    for i in range(v):
        st = oracle_forward(view_inputs(scene, cams, i, H, W, bg=(0.2, 0.3, 0.4)))
        # (the 4x4 inverses of the framing run on the GPU here and on the CPU for the oracle: ulp-level
        # different matrices, so this comparison is to the north_star tolerance, not bit-exact)
        assert np.abs(c1[i].detach().cpu().numpy() - st["color"]).max() <= ATOL_PIXEL
    dec = DecoderSplattingCUDA((0.2, 0.3, 0.4)).to(dev)
    gs = Gaussians(*(g[k][None] for k in ("means", "covariances", "harmonics", "opacities")))
    out = dec(gs, cam["extrinsics"][None], cam["intrinsics"][None], cam["near"][None], cam["far"][None], (H, W),
              depth_mode="depth")
    assert out.color.shape == (1, v, 3, H, W) and out.depth.shape == (1, v, H, W)
    assert torch.equal(out.color[0], c1) and torch.equal(out.depth[0], d1[:, 0] / 2)


def test_frame_views_matches_reference_framing(hip_device):
    """fs_frame_views against the reference's own framing (tests/golden/framing.npz, generated from
    cuda_splatting.py / projection.py by make_golden.py)."""
    import os
    from freesplat_amd.decoder import frame_views
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "framing.npz"))
    g = {k: torch.from_numpy(z[k]).to(hip_device) for k in z.files}
    campos, scale, tanfov, view, full = frame_views(g["extrinsics"], g["intrinsics"], g["near"], g["far"], True)
    np.testing.assert_allclose(tanfov.cpu().numpy(), z["tan"], rtol=2e-6)
    np.testing.assert_allclose(view.cpu().numpy(), z["view"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(full.cpu().numpy(), z["full"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(campos.cpu().numpy(), z["campos"], rtol=1e-6)
    np.testing.assert_allclose(scale.cpu().numpy(), 1.0 / z["near"], rtol=1e-7)
    _, s1, _, _, _ = frame_views(g["extrinsics"], g["intrinsics"], g["near"], g["far"], False)
    assert (s1 == 1).all()


def test_splatter_script_scenario(hip_device):
    """The scenario of the reference's own smoke script (src/scripts/test_splatter.py): ONE unit Gaussian at the
    origin, opacity 1, only the degree-2 coefficients of the red channel set (= 10), cameras spinning around it at
    radius 10 with fx = fy = 0.5, near 0.1 / far 20 -- restricted to the degrees the rasterizer supports (the script
    asks for 25 coefficients; the extension evaluates at most degree 3).  Product (render_cuda, the script's call) vs
    the oracle, plus the structure the script's comments describe (green / blue carry no harmonics)."""
    from freesplat_amd import synthetic
    from freesplat_amd.decoder import render_cuda
    from util_framing import _frame
    from oracle import raster_oracle as ro
    H = W = 128
    frames = 6
    ang = np.linspace(0.0, 2 * np.pi, frames, endpoint=False)
    c2w = np.stack([synthetic._look_at_c2w(np.array([10 * np.sin(a), 0.0, -10 * np.cos(a)]), np.zeros(3)) for a in ang])
    E = torch.from_numpy(c2w.astype(np.float32))
    K = torch.tensor([[0.5, 0, 0.5], [0, 0.5, 0.5], [0, 0, 1.0]]).expand(frames, 3, 3).contiguous()
    near, far = torch.full((frames,), 0.1), torch.full((frames,), 20.0)
    q = torch.linalg.qr(torch.randn(3, 3, generator=torch.Generator().manual_seed(4)))[0]
    cov = (q @ torch.eye(3) @ q.T)[None]                      # rotation @ diag(1) @ rotation^T
    means, opac = torch.zeros(1, 3), torch.ones(1)
    sh = torch.zeros(1, 3, 9)
    sh[:, 0, 4:9] = 10.0
    dev = hip_device
    rep = lambda x: x[None].expand(frames, *x.shape).to(dev)
    with torch.no_grad():
        color, depth = render_cuda(E.to(dev), K.to(dev), near.to(dev), far.to(dev), (H, W), torch.zeros(frames, 3, device=dev),
                                   rep(means), rep(cov), rep(sh), rep(opac))
    color = color.cpu().numpy()
    extr, scale, tx, ty, view, full = _frame(E, K, near, far, True)
    r, c = torch.triu_indices(3, 3)
    for i in range(frames):
        s = scale[i]
        st = ro.forward(H, W, float(tx[i]), float(ty[i]), np.zeros(3, np.float32), view[i].numpy(), full[i].numpy(), 2,
                        extr[i, :3, 3].numpy(), (means * s).numpy(), (cov * s * s)[:, r, c].numpy(), opac.numpy(),
                        shs=sh.transpose(-1, -2).contiguous().numpy())
        assert np.abs(color[i] - st["color"]).max() <= ATOL_PIXEL
        a = st["alpha"]
        assert a.max() > 0.9 and a[0, 0] == 0.0                               # an opaque blob in the middle
        np.testing.assert_allclose(color[i, 1], 0.5 * a, atol=1e-5)           # green, blue: DC 0 -> 0.5, no harmonics
        np.testing.assert_allclose(color[i, 2], 0.5 * a, atol=1e-5)
    assert color[0, 0].max() > 1.0 and color[2, 0].max() == 0.0   # red follows the degree-2 lobes as the camera turns


def test_cpu_tensor_raises(hip_device):
    from freesplat_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    scene, cams = small_scene(N=10, H=16, W=16, seed=1)
    vi = view_inputs(scene, cams, 0, 16, 16)
    s = GaussianRasterizationSettings(16, 16, vi["tanfovx"], vi["tanfovy"], vi["bg"], 1.0, vi["viewmatrix"],
                                      vi["projmatrix"], 2, vi["campos"], False, False)
    with pytest.raises(RuntimeError):
        GaussianRasterizer(s)(means3D=vi["means3D"], means2D=None, shs=vi["shs"], opacities=vi["opacities"],
                              cov3D_precomp=vi["cov3D"])
    with pytest.raises(Exception):
        GaussianRasterizer(s)(means3D=vi["means3D"].to(hip_device), means2D=None, opacities=vi["opacities"],
                              cov3D_precomp=vi["cov3D"])  # neither shs nor colors


@pytest.mark.parametrize("workload", ["c2_640x480_300k", "c3_968x1296_1M", "c3_closeup_968x1296_1M"])
def test_full_size_parity_and_properties(hip_device, workload):
    """BASELINE.json full sizes: pixel parity + PSNR vs the oracle, and size-independent properties
    (tile lists sorted, colour linear in the SH DC term).  The close-up workload (round 5: every tile list beyond the LDS sort's
    2 048 keys, 19.6 M culled entries) takes the two-level distribution sort on every tile."""
    H, W, N = synthetic.WORKLOADS[workload]
    scene = synthetic.workload_scene(workload)
    cams = synthetic.target_cameras(2)
    vi = view_inputs(scene, cams, 0, H, W)
    st, (color, radii, depth, alpha), _ = _check_forward(vi, hip_device)
    mse = float(((color.cpu().numpy().clip(0, 1) - st["color"].clip(0, 1)) ** 2).mean())
    psnr = float("inf") if mse == 0 else -10 * np.log10(mse)
    assert psnr > 80.0
    dbg, _, _ = _internal_state(vi, hip_device)
    frac = _check_lists(dbg, st, W, H, True, sample=20000)
    print(f"{workload}: instances kept by tile culling: {frac:.3f} of {st['num_rendered']}")
    # sortedness of every tile list by (depth bits, id)
    off, pl = dbg["offsets"].astype(np.int64), dbg["point_list"].astype(np.int64)
    key = (dbg["rec"][:, 7].view(np.uint32).astype(np.int64)[pl] << 32) | pl
    tile_of = np.repeat(np.arange(len(off) - 1), np.diff(off))
    same = tile_of[1:] == tile_of[:-1]
    assert (np.diff(key)[same] > 0).all()


@pytest.mark.parametrize("workload", ["c2_640x480_300k", "c3_968x1296_1M",
                                      pytest.param("c3_968x1296_1M", marks=pytest.mark.fast_exp, id="c3_fast_exp")])
def test_full_size_forward_backward_gradients(hip_device, workload):
    """BASELINE.json config 3 is a fwd+bwd training step at 968x1296 / 1.0 M Gaussians (config 2's size as well):
    ONE view, colour and depth cotangents, all five gradients (means3D, cov3D, SH, opacities, screen-space
    means2D) against the oracle's double-accumulated backward -- 2e-4 of each gradient's max-abs."""
    from oracle import raster_oracle as ro
    H, W, N = synthetic.WORKLOADS[workload]
    scene = synthetic.make_scene(N)
    cams = synthetic.target_cameras(2)
    vi = view_inputs(scene, cams, 1, H, W, bg=(0.2, 0.1, 0.3))
    st = oracle_forward(vi)
    rng = np.random.default_rng(5)
    g_color = rng.normal(size=(3, H, W)).astype(np.float32)
    g_depth = (0.25 * rng.normal(size=(H, W))).astype(np.float32)
    ref = ro.backward(st, g_color, g_depth)
    (color, radii, depth, alpha), leaves = hip_forward(vi, hip_device, requires_grad=True)
    from freesplat_amd import rasterizer as R
    if R.FAST_EXP:     # opt-in hardware exp: same accept / reject decisions (test_fast_exp_mode_quantified); gradients below
        d = np.abs(color.detach().cpu().numpy() - st["color"]).max(axis=0)
        assert int((d > ATOL_PIXEL).sum()) <= 1e-6 * H * W and d.max() <= 2e-4
    else:              # the forward half: bit-exact
        np.testing.assert_array_equal(color.detach().cpu().numpy(), st["color"])
        np.testing.assert_array_equal(depth.detach().cpu().numpy(), st["depth"])
    loss = (color * torch.from_numpy(g_color).to(hip_device)).sum() + (depth * torch.from_numpy(g_depth).to(hip_device)).sum()
    loss.backward()
    worst = {}
    for name, key in (("means3D", "means3D"), ("cov3D", "cov3D"), ("shs", "shs"), ("opacities", "opacities")):
        got = leaves[name].grad.cpu().numpy().reshape(ref[key].shape)
        assert np.isfinite(got).all(), name
        worst[name] = float(np.abs(got - ref[key]).max() / (np.abs(ref[key]).max() + 1e-20))
    m2 = leaves["means2D"].grad.cpu().numpy()
    worst["means2D"] = float(np.abs(m2[:, :2] - ref["means2D"]).max() / (np.abs(ref["means2D"]).max() + 1e-20))
    print(f"{workload} fwd+bwd: gradient error / max-abs = {worst}")
    assert (m2[:, 2] == 0).all()
    assert max(worst.values()) < 2e-4, worst      # (the same bar in the hardware-exp mode: its decisions are the exact mode's)
    # culled Gaussians get exactly zero gradient
    dead = st["radii"] == 0
    assert dead.any() and not leaves["means3D"].grad.cpu().numpy()[dead].any()


@pytest.mark.parametrize("workload", ["c2_640x480_300k", "c3_968x1296_1M"])
def test_exp_contract_sensitivity(hip_device, workload):
    """The forward is bit-exact against an oracle that shares the kernels' private exp() (Cody-Waite + polynomial).
    The reference's CUDA extension uses its own expf, which no CPU can replay; what CAN be measured is how much of
    the image hangs on that choice: the same oracle with libm expf() in the blend loop (differs from the contract
    exp by <= 1 ulp) against the HIP image.  The north_star's bar (<= 1e-4 abs per pixel) must hold across that
    swap too, otherwise bit-exactness against the contract oracle would say nothing about the real reference."""
    from oracle import raster_oracle as ro
    H, W, N = synthetic.WORKLOADS[workload]
    scene = synthetic.make_scene(N)
    cams = synthetic.target_cameras(2)
    vi = view_inputs(scene, cams, 0, H, W)
    (color, _, depth, _), _ = hip_forward(vi, hip_device)
    c = color.cpu().numpy()
    try:
        ro.set_exp_mode(True)
        st = oracle_forward(vi)
    finally:
        ro.set_exp_mode(False)
    diff = np.abs(c - st["color"]).max(axis=0)
    n_bad = int((diff > ATOL_PIXEL).sum())
    mse = float(((c.clip(0, 1) - st["color"].clip(0, 1)) ** 2).mean())
    psnr = float("inf") if mse == 0 else -10 * np.log10(mse)
    print(f"{workload}: HIP (contract exp) vs oracle with libm expf: max-abs {diff.max():.3e}, pixels > 1e-4: {n_bad} "
          f"of {H * W}, PSNR {psnr:.1f} dB")
    assert n_bad <= 1e-5 * H * W and diff.max() <= 5e-3 and psnr > 100.0


def test_render_views_hipgraph_capture_and_replay(hip_device):
    """The C ABI allocates nothing and never syncs, so a whole decoder call -- framing + 5 kernels per view, views
    alternating over two forked side streams -- records into ONE hipGraph (torch.cuda.CUDAGraph on ROCm) and replays
    with new Gaussian values written into the captured input buffers.  (What small workloads such as config 2 need when
    the per-view launch train, not the GPU, is the limit.)"""
    from freesplat_amd.decoder import check_deferred, render_views
    H, W, v = 96, 128, 4
    dev = hip_device
    scene, cams = small_scene(N=6000, H=H, W=W, seed=31, n_views=v)
    g = {k: scene[k].to(dev) for k in ("means", "covariances", "harmonics", "opacities")}
    cam = {k: t.to(dev) for k, t in cams.items()}
    bg = torch.zeros(v, 3, device=dev)

    def call():
        return render_views(cam["extrinsics"], cam["intrinsics"], cam["near"], cam["far"], (H, W), bg, g["means"],
                            g["covariances"], g["harmonics"], g["opacities"], check="deferred")

    with torch.no_grad():
        call(); check_deferred()                       # warm-up outside capture (side streams, caches)
        torch.cuda.synchronize()
        side = torch.cuda.Stream(device=dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                color, depth = call()
        from freesplat_amd import decoder as _D
        _D._pending_checks.clear()                     # (the captured call's counters are checked after each replay below)
        for trial in range(2):
            if trial:                                  # new scene values into the SAME buffers, then replay
                scene2, _ = small_scene(N=6000, H=H, W=W, seed=77, n_views=v)
                for k in g:
                    g[k].copy_(scene2[k].to(dev))
            graph.replay()
            torch.cuda.synchronize()
            got_c, got_d = color.clone(), depth.clone()
            ref_c, ref_d = call(); check_deferred()
            assert torch.equal(got_c, ref_c) and torch.equal(got_d, ref_d), f"replay {trial}"
        assert got_c.abs().max() > 0


@pytest.mark.fast_exp
@pytest.mark.parametrize("H,W,N,seed,workload", [(64, 80, 600, 7, None), (256, 256, 20000, 5, None),
                                                 (0, 0, 0, 0, "c2_640x480_300k"), (0, 0, 0, 0, "c3_968x1296_1M")])
def test_fast_exp_mode_quantified(hip_device, monkeypatch, H, W, N, seed, workload):
    """The opt-in hardware exp (FS_RASTER_FAST_EXP) against the oracle: everything the exp does not touch is still
    IDENTICAL -- radii, tile ranges, depth-sorted id lists.  The alpha >= 1/255 decisions are those of the exact mode too
    (a step with an alpha inside the +-16 ulp guard band of the threshold is re-evaluated with the contract exp), so the
    image agrees to ~1e-6; the one discontinuity left is the T >= 1e-4 termination, whose flip changes a pixel by less
    than its remaining transmittance (1e-4 x colour).  Bars: at most one pixel per million above the north_star's 1e-4,
    none above 2e-4, median <= 1e-6, PSNR > 100 dB."""
    _set_cull(monkeypatch, False)
    if workload:
        H, W, N = synthetic.WORKLOADS[workload]
        scene, cams = synthetic.make_scene(N), synthetic.target_cameras(2)
    else:
        scene, cams = small_scene(N=N, H=H, W=W, seed=seed)
    vi = view_inputs(scene, cams, 1, H, W, bg=(0.1, 0.2, 0.3))
    from freesplat_amd import rasterizer as R
    assert R.FAST_EXP
    st = oracle_forward(vi)
    (color, radii, depth, alpha), _ = hip_forward(vi, hip_device)
    np.testing.assert_array_equal(radii.cpu().numpy(), st["radii"])
    c = color.cpu().numpy()
    d = np.abs(c - st["color"]).max(axis=0)
    n_bad = int((d > ATOL_PIXEL).sum())
    mse = float(((c.clip(0, 1) - st["color"].clip(0, 1)) ** 2).mean())
    psnr = float("inf") if mse == 0 else -10 * np.log10(mse)
    print(f"fast exp {workload or (H, W, N)}: max-abs {d.max():.2e}, median {np.median(d):.1e}, pixels > 1e-4: {n_bad} of {H * W}, "
          f"PSNR {psnr:.1f} dB")
    assert n_bad <= max(1, int(1e-6 * H * W)) and d.max() <= 2e-4 and psnr > 100.0 and np.median(d) <= 1e-6
    dbg, _, _ = _internal_state(vi, hip_device)
    np.testing.assert_array_equal(dbg["offsets"][1:], st["ranges"][:, 1])
    np.testing.assert_array_equal(dbg["point_list"], st["point_list"])
    assert float((dbg["n_contrib"] != st["n_contrib"]).mean()) < 1e-4      # termination flips only


@pytest.mark.parametrize("N", [2300, 3000, 4000, 5200, 9500, 15000])
def test_dense_tiles_long_lists(hip_device, monkeypatch, N):
    """Every Gaussian of the scene lands on the same 2x2 tiles: per-tile lists of ~N entries drive the sort paths
    beyond the LDS bucket sort (2048 < n <= 4096: register bitonic networks, two-run and full; n > 4096: the in-place
    global-memory network) and the blend through 40-80 batches per quadrant.  Bit-exact images, identical lists."""
    _set_cull(monkeypatch, False)
    H = W = 32
    scene, cams = small_scene(N=N, H=H, W=W, seed=44)
    # pull every Gaussian towards the optical axis so that its 3-sigma square covers the whole 32x32 image
    scene["means"][:, :2] = scene["means"][:, :2] * 0.02
    scene["covariances"] = scene["covariances"] * 400.0
    scene["opacities"] = scene["opacities"] * 0.004           # keep the pixels unsaturated: deep lists are blended
    vi = view_inputs(scene, cams, 0, H, W, bg=(0.05, 0.1, 0.15))
    st, _, _ = _check_forward(vi, hip_device)
    n_tile = (st["ranges"][:, 1] - st["ranges"][:, 0])
    assert n_tile.max() > min(N, 4096) * 0.55, n_tile
    dbg, _, _ = _internal_state(vi, hip_device)
    np.testing.assert_array_equal(dbg["point_list"], st["point_list"])
    np.testing.assert_array_equal(dbg["n_contrib"], st["n_contrib"])
    assert st["n_contrib"].max() > 0.4 * n_tile.max()           # deep lists really are walked


@pytest.mark.parametrize("cull", [False, True])
@pytest.mark.parametrize("H,W,N,scale", [(256, 256, 400, 900.0), (1040, 1040, 600, 2500.0), (17, 33, 300, 30.0)])
def test_large_rects_and_odd_sizes(hip_device, monkeypatch, H, W, N, scale, cull):
    """Gaussians whose 3-sigma squares span many tiles (rects of > 16 tiles: the row-band path of preprocess / emit
    instead of the packed 64-bit masks; at 1040x1040 = 65x65 tiles a workgroup's bounding box exceeds the 4096-counter
    LDS window and binning falls back to direct global atomics), and an image that is no multiple of the tile size."""
    _set_cull(monkeypatch, cull)
    scene, cams = small_scene(N=N, H=H, W=W, seed=61)
    scene["covariances"] = scene["covariances"] * scale
    scene["opacities"] = scene["opacities"] * 0.05
    vi = view_inputs(scene, cams, 0, H, W, bg=(0.3, 0.2, 0.1))
    st, _, _ = _check_forward(vi, hip_device)
    area = (st["rect"][:, 2] - st["rect"][:, 0]) * (st["rect"][:, 3] - st["rect"][:, 1])
    if H >= 256:
        assert (area > 16).mean() > 0.3 and area.max() >= 64, (area.max(), (area > 16).mean())
    dbg, _, _ = _internal_state(vi, hip_device)
    _check_lists(dbg, st, W, H, cull, sample=5000)
    np.testing.assert_array_equal(dbg["final_T"], st["final_T"])


def test_backward_large_gaussians_and_deep_lists(hip_device):
    """Backward where every pixel has hundreds of contributors and the Gaussians span many tiles (the per-Gaussian
    atomics of one Gaussian come from ~100 quadrants): all gradients vs the oracle's double accumulation."""
    from oracle import raster_oracle as ro
    H, W, N = 96, 128, 1500
    scene, cams = small_scene(N=N, H=H, W=W, seed=71)
    scene["covariances"] = scene["covariances"] * 300.0
    scene["opacities"] = scene["opacities"] * 0.01
    vi = view_inputs(scene, cams, 1, H, W, bg=(0.2, 0.4, 0.6))
    st = oracle_forward(vi)
    assert st["n_contrib"].mean() > 150
    rng = np.random.default_rng(2)
    g_color = rng.normal(size=(3, H, W)).astype(np.float32)
    g_depth = rng.normal(size=(H, W)).astype(np.float32)
    ref = ro.backward(st, g_color, g_depth)
    (color, radii, depth, alpha), leaves = hip_forward(vi, hip_device, requires_grad=True)
    ((color * torch.from_numpy(g_color).to(hip_device)).sum() + (depth * torch.from_numpy(g_depth).to(hip_device)).sum()).backward()
    for name in ("means3D", "cov3D", "shs", "opacities"):
        got = leaves[name].grad.cpu().numpy().reshape(ref[name].shape)
        err = np.abs(got - ref[name]).max() / (np.abs(ref[name]).max() + 1e-20)
        assert err < 2e-4, f"{name}: {err}"


def test_inference_forward_keeps_no_backward_state(hip_device):
    """A forward without any input requiring grad runs the blend without the contributor count
    (FS_RASTER_NO_BACKWARD_STATE): same image bits as the tracking kernel, and the C ABI refuses a backward on it."""
    from freesplat_amd import _lib, rasterizer as R
    scene, cams = small_scene(N=3000, H=80, W=96, seed=12)
    vi = view_inputs(scene, cams, 0, 80, 96, bg=(0.2, 0.3, 0.1))
    (c_inf, _, d_inf, a_inf), _ = hip_forward(vi, hip_device)                       # no grad: inference kernel
    (c_trk, _, d_trk, a_trk), _ = hip_forward(vi, hip_device, requires_grad=True)   # tracking kernel
    assert c_inf.grad_fn is None and c_trk.grad_fn is not None
    assert not (c_trk.grad_fn.rs.dims.flags & _lib.RASTER_NO_BACKWARD_STATE)
    assert torch.equal(c_inf, c_trk.detach()) and torch.equal(d_inf, d_trk.detach()) and torch.equal(a_inf, a_trk.detach())
    # the same forward launched by hand with the flag, then a backward on its state
    d = lambda t: t.to(hip_device)
    s = R.GaussianRasterizationSettings(vi["H"], vi["W"], vi["tanfovx"], vi["tanfovy"], d(vi["bg"]), 1.0, d(vi["viewmatrix"]),
                                        d(vi["projmatrix"]), vi["sh_degree"], d(vi["campos"]), False, False)
    dims = R.make_dims(vi["means3D"].shape[0], vi["shs"].shape[1], s, inference=True)
    assert dims.flags & _lib.RASTER_NO_BACKWARD_STATE
    rs, color, _, _ = R.rasterize_forward_checked(dims, d(vi["means3D"]), d(vi["cov3D"]), d(vi["shs"]), None, d(vi["opacities"]),
                                                  d(vi["bg"]), d(vi["viewmatrix"]), d(vi["projmatrix"]), d(vi["campos"]))
    assert torch.equal(color, c_inf)
    with pytest.raises(_lib.FreeSplatHipError):
        R.rasterize_backward(rs, d(vi["means3D"]), d(vi["cov3D"]), d(vi["shs"]), None, d(vi["opacities"]), torch.ones_like(color), None)
