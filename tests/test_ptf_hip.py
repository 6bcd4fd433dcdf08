"""GPU parity of the PTF path: fs_ptf_match (index work -> bit-exact vs the oracle) and the whole
fold against the reference's golden outputs and the oracle (fp32 tolerance 1e-4, identical ORDER)."""
import os
import sys

import numpy as np
import pytest
import torch

from ptf_torch_ref import fuse_gaussians_torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
pytestmark = pytest.mark.gpu


_GRU_KEYS = [f"{m}.{l}.{t}" for m in ("mlp_r", "mlp_z", "mlp_n") for l in ("0", "2") for t in ("weight", "bias")]   # ptf._gru_params order


def _oracle_pe(positions, freqs):
    from oracle.ptf_oracle import positional_encoding      # the reference-pinned restatement (encoder_freesplat.py:62-77)
    return positional_encoding(positions, freqs)


def _oracle_gru_rows(params, cat):
    """oracle/ptf_oracle.py:gru (networks.py:201-214, pinned by the reference's golden folds) on rows
    cat = [hid(64) | he(24) | x(64) | xe(24)] with the parameters in ptf._gru_params order."""
    from oracle.ptf_oracle import gru
    return gru(dict(zip(_GRU_KEYS, params)), cat[:, 88:152], cat[:, :64], cat[:, 152:], cat[:, 64:88])


def _load(name):
    z = np.load(os.path.join(HERE, "golden", name))
    g = {k: torch.from_numpy(z[k]) if z[k].ndim else z[k].item() for k in z.files}
    gru = {k[len("gru__"):].replace("__", "."): v for k, v in g.items() if k.startswith("gru__")}
    return g, gru


def _scene(V, h, w, seed, noise=0.02):
    import inputs
    from oracle import ptf_oracle as po
    E, Kn, depths, lat, dens, wts = inputs.ptf_inputs(V, h, w, seed=seed)
    depths = 2.0 + noise * torch.randn(V, 1, h, w, generator=torch.Generator().manual_seed(seed))
    # unproject like gaussian_adapter.py:36-79 (integer pixel coords, no +0.5)
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    coords = []
    for i in range(V):
        K = Kn[i].clone(); K[0] *= w; K[1] *= h
        z = depths[i, 0]
        pc = torch.stack([(xs - K[0, 2]) / K[0, 0] * z, (ys - K[1, 2]) / K[1, 1] * z, z, torch.ones_like(z)], -1).reshape(-1, 4)
        coords.append((pc @ E[i].T)[:, :3])
    coords = torch.stack(coords)[None, :, :, None, None, :]
    return E, Kn, depths, lat, dens, wts, coords


@pytest.mark.parametrize("h,w,M,seed", [(24, 32, 768, 1), (48, 64, 10000, 2), (384, 512, 196608 * 2, 3), (7, 9, 5, 4),
                                         (968, 1296, 1254528 * 2, 5)])     # (the last: BASELINE config 3's image, 2 views of state)
def test_match_bit_exact_vs_oracle(hip_device, h, w, M, seed):
    from oracle import ptf_oracle as po
    from freesplat_amd.ptf import match_view
    rng = np.random.default_rng(seed)
    P = h * w
    fx, fy, cx, cy = 0.9 * w, 1.2 * h, 0.49 * w, 0.51 * h
    # points spread over (and beyond) the frustum at depth ~2, many pairs sharing pixels, some behind
    u = rng.uniform(-0.1 * w, 1.1 * w, M); v = rng.uniform(-0.1 * h, 1.1 * h, M)
    z = 2.0 + 0.05 * rng.normal(size=M)
    z[rng.random(M) < 0.02] *= -1
    xyz = np.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], -1).astype(np.float32)
    xyz[: M // 10] = xyz[M // 10: 2 * (M // 10)][: M // 10]          # exact duplicates -> z ties
    ang = 0.03
    w2c = np.array([[np.cos(ang), 0, np.sin(ang), 0.01], [0, 1, 0, -0.02], [-np.sin(ang), 0, np.cos(ang), 0.03],
                    [0, 0, 0, 1]], np.float32)
    kpix = np.array([fx, fy, cx, cy], np.float32)
    depth_i = (2.0 + 0.05 * rng.normal(size=P)).astype(np.float32)
    ref = po.match_step(xyz, w2c, kpix, depth_i, h, w)
    t = lambda a: torch.from_numpy(a).to(hip_device)
    got = match_view(t(xyz), t(w2c), t(kpix), t(depth_i), h, w)
    for a, b, name in zip(got, ref, ("keep", "fuse", "fuse_pix", "append")):
        np.testing.assert_array_equal(a.cpu().numpy(), b, err_msg=name)
    assert len(ref[1]) > 0 or M < 10


def test_match_empty_state(hip_device):
    from freesplat_amd.ptf import match_view
    h, w = 8, 8
    d = torch.full((h * w,), 2.0, device=hip_device)
    keep, fuse, fpix, app = match_view(torch.zeros(0, 3, device=hip_device), torch.eye(4, device=hip_device),
                                       torch.tensor([4.0, 4.0, 4.0, 4.0], device=hip_device), d, h, w)
    assert keep.numel() == 0 and fuse.numel() == 0 and torch.equal(app.cpu(), torch.arange(h * w))


def _run_fold(g, gru_params, dev):
    from freesplat_amd.ptf import PixelwiseTripletFusion
    m = PixelwiseTripletFusion()
    m.gru.load_state_dict(gru_params, strict=True)
    m = m.to(dev)
    d = lambda t: t.to(dev)
    return m, m.fuse_gaussians([d(g["latents"])], [d(g["coords"])], d(g["densities"]), d(g["weights"]), d(g["depths"]),
                               d(g["extrinsics"])[None], d(g["intrinsics"])[None], (int(g["h"]), int(g["w"])))


@pytest.mark.parametrize("grad", [False, True])
@pytest.mark.parametrize("name", ["ptf_small.npz", "ptf_tie.npz"])
def test_fold_matches_reference_golden(hip_device, name, grad):
    """grad=False: inference (fs_ptf_fold, one library call); grad=True: the training path (_PtfFold: the same HIP
    fold step by step, state kept for its HIP backward)."""
    g, gru = _load(name)
    with torch.set_grad_enabled(grad):
        _, out = _run_fold(g, gru, hip_device)
    for got, key in zip(out, ("out_latent", "out_xyz", "out_extrinsics", "out_depths")):
        assert got.shape == g[key].shape, key
        assert (got.detach().cpu() - g[key]).abs().max().item() <= 1e-4, key


@pytest.mark.parametrize("V,h,w", [(2, 96, 128), (5, 48, 64)])
def test_fused_inference_path_equals_differentiable_path(hip_device, V, h, w):
    from freesplat_amd.ptf import PixelwiseTripletFusion
    E, Kn, depths, lat, dens, wts, coords = _scene(V, h, w, seed=60 + V)
    torch.manual_seed(2)
    m = PixelwiseTripletFusion().to(hip_device)
    d = lambda t: t.to(hip_device)
    a = ([d(lat)], [d(coords)], d(dens), d(wts), d(depths), d(E)[None], d(Kn)[None], (h, w))
    with torch.no_grad():
        fused = m.fuse_gaussians(*a)
    ref = m.fuse_gaussians(*a)   # GRU parameters require grad -> training path (HIP fold + HIP backward)
    tor = fuse_gaussians_torch(m, *a)   # op-by-op torch formulation on the HIP index lists (tests/ptf_torch_ref.py)
    assert ref[0].requires_grad and tor[0].requires_grad and not fused[0].requires_grad
    for x, y, t, name in zip(fused, ref, tor, ("latent", "xyz", "extrinsics", "depths")):
        assert x.shape == y.shape == t.shape, name
        assert torch.equal(x, y.detach()), name                      # same kernels, same bits
        assert (x - t.detach()).abs().max().item() <= 2e-5, name


@pytest.mark.parametrize("V,h,w", [(2, 96, 128), (4, 48, 64)])
def test_fold_vs_oracle_and_gradients(hip_device, V, h, w):
    """Training path: outputs vs the oracle, and the gradients of EVERY differentiable input (latents, coords,
    densities, weight embeddings, depths) and of the 12 GRU parameters, with cotangents on all four outputs,
    against (i) the oracle's CPU autograd and (ii) the torch-op formulation on the same device."""
    from oracle import ptf_oracle as po
    from freesplat_amd.ptf import PixelwiseTripletFusion
    E, Kn, depths, lat, dens, wts, coords = _scene(V, h, w, seed=40 + V)
    torch.manual_seed(9)
    m = PixelwiseTripletFusion()
    params = {k: v.detach().clone().requires_grad_(True) for k, v in m.gru.state_dict().items()}
    names = ("latents", "coords", "densities", "weights", "depths")
    cpu_in = [t.clone().requires_grad_(True) for t in (lat, coords, dens, wts, depths)]
    ref = po.fuse_gaussians(params, cpu_in[0], cpu_in[1], cpu_in[2], cpu_in[3], cpu_in[4], E[None], Kn[None], (h, w))
    gen = torch.Generator().manual_seed(1)
    cot = [torch.randn(r.shape, generator=gen) for r in ref]
    sum((r * c).sum() for r, c in zip(ref, cot)).backward()
    m = m.to(hip_device)
    d = lambda t: t.to(hip_device)

    def run(fn):
        ins = [d(t).requires_grad_(True) for t in (lat, coords, dens, wts, depths)]
        for q in m.gru.parameters():
            q.grad = None
        out = fn([ins[0]], [ins[1]], ins[2], ins[3], ins[4], d(E)[None], d(Kn)[None], (h, w))
        sum((o * d(c)).sum() for o, c in zip(out, cot)).backward()
        return out, [t.grad for t in ins], {k: q.grad.clone() for k, q in m.gru.named_parameters()}

    out, gin, gpar = run(m.fuse_gaussians)
    assert out[0].shape == ref[0].shape and out[0].shape[1] < V * h * w   # something fused
    for a, b, name in zip(out, ref, ("latent", "xyz", "extrinsics", "depths")):
        assert (a.detach().cpu() - b.detach()).abs().max().item() <= 1e-4, name
    out_t, gin_t, gpar_t = run(lambda *a_: fuse_gaussians_torch(m, *a_))
    rel = lambda a, b: (a - b).abs().max().item() / (b.abs().max().item() + 1e-20)
    for a, t, c, name in zip(gin, gin_t, cpu_in, names):
        assert a is not None and a.shape == c.grad.shape, name
        assert rel(a.cpu(), c.grad) < 1e-3, f"d {name} vs oracle autograd: {rel(a.cpu(), c.grad)}"
        assert rel(a, t) < 1e-3, f"d {name} vs torch-op path: {rel(a, t)}"
    for k in gpar:
        assert rel(gpar[k].cpu(), params[k].grad) < 2e-3, f"d gru.{k} vs oracle autograd: {rel(gpar[k].cpu(), params[k].grad)}"
        assert rel(gpar[k], gpar_t[k]) < 2e-3, f"d gru.{k} vs torch-op path"


def test_fold_backward_with_tied_winners(hip_device):
    """ptf_tie.npz (an exact z tie: two Gaussians fuse with the SAME pixel): the view-side gradients accumulate over
    both fused rows (float atomics), as autograd's index_select backward does."""
    from freesplat_amd.ptf import PixelwiseTripletFusion
    g, gru = _load("ptf_tie.npz")
    m = PixelwiseTripletFusion()
    m.gru.load_state_dict(gru, strict=True)
    m = m.to(hip_device)
    d = lambda t: t.to(hip_device)
    res = []
    for fn in (m.fuse_gaussians, lambda *a_: fuse_gaussians_torch(m, *a_)):
        ins = [d(g[k]).clone().requires_grad_(True) for k in ("latents", "coords", "densities", "weights", "depths")]
        out = fn([ins[0]], [ins[1]], ins[2], ins[3], ins[4], d(g["extrinsics"])[None], d(g["intrinsics"])[None],
                 (int(g["h"]), int(g["w"])))
        gen = torch.Generator().manual_seed(4)
        sum((o * d(torch.randn(o.shape, generator=gen))).sum() for o in out).backward()
        res.append([t.grad for t in ins])
    for a, b in zip(*res):
        assert (a - b).abs().max().item() <= 1e-3 * (b.abs().max().item() + 1e-20)


@pytest.mark.parametrize("n", [1, 31, 32, 33, 130, 5000, 16411])
def test_gru_backward_kernel_vs_autograd(hip_device, n):
    """fs_ptf_gru_backward (forward re-run + the six transposed layers on the matrix cores, ragged last group of 32) and
    fs_ptf_gru_weight_grads (dW = dY^T X over the pairs on the matrix cores: one workgroup at n <= 64, a ragged last
    workgroup, the full 256-workgroup grid at 16411) against torch autograd of the same GRU (networks.py:201-214) on
    the same rows: input-row gradient and all 12 parameter gradients within 1e-4 of each one's max-abs."""
    from freesplat_amd import ptf as P
    torch.manual_seed(100 + n)
    gru = P.GRU().to(hip_device)
    with torch.no_grad():
        for q in gru.parameters():
            q.mul_(1.5)                              # push more units through the ReLU / gate non-linearities
    cat = torch.randn(n, 176, device=hip_device)
    cat[:, 64:88] = torch.sin(cat[:, 64:88] * 3.0)   # positional-encoding-like ranges
    cat[:, 152:] = torch.cos(cat[:, 152:] * 3.0)
    g = torch.randn(n, 64, device=hip_device)
    params = P._gru_params(gru)
    dcat, grads = P.gru_backward(params, P.gru_tables(gru), P.gru_operand_stream(gru), cat, g)
    cat_ = cat.clone().requires_grad_(True)
    ps = [q.detach().clone().requires_grad_(True) for q in params]
    ref = torch.autograd.grad(_oracle_gru_rows(ps, cat_), [cat_] + ps, g)
    rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-20))
    worst = {"dcat": rel(dcat, ref[0])}
    for k, (a, b) in enumerate(zip(grads, ref[1:])):
        assert a.shape == b.shape, k
        worst[f"param{k}"] = rel(a, b)
    assert max(worst.values()) < 1e-4, worst


@pytest.mark.parametrize("n", [1, 5, 1000])
def test_gru_inputs_rows_and_their_backward(hip_device, n):
    """fs_ptf_gru_inputs (the materialised rows [hid | PE(rho_i, O) | x | PE(R, om_i)] the training backward re-gathers,
    encoder_freesplat.py:485-486; every lane of a pair's group computes some of the 24 (sin, cos) pairs) against the torch
    formulation, values up to the hundreds (a long fold's accumulated densities); fs_ptf_gru_inputs_backward (lanes share
    the encodings' derivative, contiguous atomics into the view's latent gradient) against torch autograd of the same
    rows, with two pairs TIED on one pixel."""
    from freesplat_amd import _lib, ptf as P
    L, p = _lib.lib(), _lib.ptr
    g = torch.Generator().manual_seed(n)
    M, Pn = 3 * n + 2, 2 * n + 3
    G = torch.randn(M, 64, generator=g).to(hip_device)
    R = (torch.rand(M, generator=g) * 300).to(hip_device)
    O = (torch.rand(M, generator=g) * 40).to(hip_device)
    lat = torch.randn(Pn, 64, generator=g).to(hip_device)
    rho = torch.rand(Pn, generator=g).to(hip_device)
    om = (torch.rand(Pn, generator=g) * 3).to(hip_device)
    fuse = torch.randperm(M, generator=g)[:n].sort().values.to(hip_device)
    fpix = torch.randperm(Pn, generator=g)[:n].to(hip_device)
    if n > 1:
        fpix[1] = fpix[0]                                    # z-tied Gaussians fused with the same pixel
    cat = torch.empty(n, 176, device=hip_device)
    _lib.check(L.fs_ptf_gru_inputs(n, p(fuse), p(fpix), p(G), p(R), p(O), p(lat), p(rho), p(om), p(cat),
                                   _lib.current_stream()), "fs_ptf_gru_inputs")

    def rows(G, R, O, lat, rho, om):
        he = _oracle_pe(torch.stack([rho[fpix], O[fuse]], -1).double(), 6)
        xe = _oracle_pe(torch.stack([R[fuse], om[fpix]], -1).double(), 6)
        return torch.cat([G[fuse].double(), he, lat[fpix].double(), xe], -1)
    ins = [t.detach().clone().requires_grad_(True) for t in (G, R, O, lat, rho, om)]
    ref = rows(*ins)
    assert (cat.double() - ref).abs().max().item() < 5e-6
    dcat = torch.randn(n, 176, generator=g).to(hip_device)
    gG, gR, gO, gl, gr, go = torch.autograd.grad(ref, ins, dcat.double())
    # the kernel STORES g_G rows fuse_idx, ADDS to g_R / g_O (what fs_ptf_write_state_backward stored) and accumulates the
    # view's gradients
    hG = torch.full((M, 64), 7.0, device=hip_device)
    hR, hO = torch.ones(M, device=hip_device), torch.ones(M, device=hip_device)
    hl, hr, ho = torch.zeros(Pn, 64, device=hip_device), torch.zeros(Pn, device=hip_device), torch.zeros(Pn, device=hip_device)
    _lib.check(L.fs_ptf_gru_inputs_backward(n, p(fuse), p(fpix), p(R), p(O), p(rho), p(om), p(dcat), p(hG), p(hR), p(hO),
                                            p(hl), p(hr), p(ho), _lib.current_stream()), "fs_ptf_gru_inputs_backward")
    rel = lambda a, b: float((a.double() - b).abs().max() / (b.abs().max() + 1e-20))
    assert torch.equal(hG[fuse], dcat[:, :64]) and rel(hl, gl) < 1e-6
    keep = torch.ones(M, dtype=torch.bool, device=hip_device)
    keep[fuse] = False
    assert bool((hG[keep] == 7.0).all())
    assert rel(hR - 1, gR) < 2e-5 and rel(hO - 1, gO) < 2e-5 and rel(hr, gr) < 2e-5 and rel(ho, go) < 2e-5


def test_gru_weight_grads_accumulate_in_a_fixed_order(hip_device):
    """fs_ptf_gru_weight_grads ADDS to `grads` (the fold steps of a scene accumulate in one buffer) and sums the
    workgroups' partials in a fixed order: two runs on the same rows give the same bits, a second call into the same
    buffer doubles every entry exactly."""
    from freesplat_amd import ptf as P
    torch.manual_seed(7)
    gru = P.GRU().to(hip_device)
    n = 9000
    cat = torch.randn(n, 176, device=hip_device)
    g = torch.randn(n, 64, device=hip_device)
    params, tab, stream = P._gru_params(gru), P.gru_tables(gru), P.gru_operand_stream(gru)
    from freesplat_amd import _lib
    flat = torch.zeros(_lib.lib().fs_ptf_gru_grad_floats(), device=hip_device)
    d1, g1 = P.gru_backward(params, tab, stream, cat, g, flat)
    once = flat.clone()
    d2, g2 = P.gru_backward(params, tab, stream, cat, g)                  # fresh zeroed buffer
    assert torch.equal(torch.cat([t.reshape(-1) for t in g2]), once) and torch.equal(d1, d2)
    P.gru_backward(params, tab, stream, cat, g, flat)                     # second step into the same buffer
    assert torch.equal(flat, 2 * once)
    assert [tuple(t.shape) for t in g1] == [tuple(q.shape) for q in params]


@pytest.mark.parametrize("n", [1, 33, 4097])
def test_gru_forward_kernel_on_materialised_rows(hip_device, n):
    """fs_ptf_gru_forward (the GRU on caller-built rows [hid | he | x | xe]; the fold itself uses the gathering variant of
    the same kernel) against the torch GRU (networks.py:201-214): 1e-5 abs on O(1) outputs."""
    from freesplat_amd import _lib, ptf as P
    torch.manual_seed(7 + n)
    gru = P.GRU().to(hip_device)
    cat = torch.randn(n, 176, device=hip_device)
    fused = torch.empty(n, 64, device=hip_device)
    tab = P.gru_tables(gru)
    _lib.check(_lib.lib().fs_ptf_gru_forward(n, _lib.ptr(cat), _lib.ptr(tab), _lib.ptr(fused), _lib.current_stream()),
               "fs_ptf_gru_forward")
    with torch.no_grad():
        ref = _oracle_gru_rows(P._gru_params(gru), cat)
    assert (fused - ref).abs().max().item() < 1e-5


def test_invert_4x4_vs_float64_inverse(hip_device):
    """fs_invert_4x4 (ptf.world_to_camera: double precision inside, rounded once) directly against torch's float64 inverse:
    camera poses, random well-conditioned matrices and near-singular ones agree to the rounding of the float32 result (the
    full-size fold tests feed the oracle THIS kernel's matrices, so an error here would cancel out there: ADVICE r3); a
    singular matrix gives non-finite entries instead of the reference's `.inverse()` exception (ptf.py documents it)."""
    import inputs
    from freesplat_amd.ptf import world_to_camera
    g = torch.Generator().manual_seed(7)
    poses = inputs.cameras(12, 96, 128, baseline=0.4, seed=3)[0]
    rnd = torch.randn(64, 4, 4, generator=g) + 2.0 * torch.eye(4)
    near = torch.eye(4).repeat(8, 1, 1)
    near[:, 2] = near[:, 1] * (1 + 1e-3 * torch.arange(1, 9).float())[:, None] + 1e-3 * torch.randn(8, 4, generator=g)   # cond ~ 1e3
    for name, M in (("poses", poses), ("random", rnd), ("near-singular", near)):
        want = torch.linalg.inv(M.double())
        got = world_to_camera(M.to(hip_device)).view(-1, 4, 4).cpu().double()
        # the float32 INPUT is exact; the only error is the final rounding of each entry (half an ulp), plus the double
        # solve's own error amplified by the condition number (negligible here)
        tol = 2.0 ** -23 * want.abs().amax(dim=(1, 2), keepdim=True)
        assert bool(((got - want).abs() <= tol).all()), (name, float(((got - want).abs() / tol).max()))
    sing = torch.eye(4)[None].clone()
    sing[0, 3] = sing[0, 2]
    assert not bool(torch.isfinite(world_to_camera(sing.to(hip_device))).all())


def test_fold_backward_matches_reference_gradients(hip_device):
    """The HIP fold's backward (_PtfFold: fs_ptf_write_state_backward, fs_ptf_gru_backward, fs_ptf_gru_inputs_backward)
    against gradients computed THROUGH THE REFERENCE'S OWN fuse_gaussians + GRU (tests/golden/ptf_small_grads.npz,
    make_golden.gen_backward): latents, coordinates, densities, weights and the 12 GRU tensors."""
    from freesplat_amd.ptf import PixelwiseTripletFusion
    g, gru = _load("ptf_small.npz")
    z = np.load(os.path.join(HERE, "golden", "ptf_small_grads.npz"))
    gg = {k: torch.from_numpy(z[k]) for k in z.files}
    m = PixelwiseTripletFusion()
    m.gru.load_state_dict(gru, strict=True)
    m = m.to(hip_device)
    d = lambda t: t.to(hip_device)
    leaves = [d(g[k]).requires_grad_(True) for k in ("latents", "coords", "densities", "weights")]
    out = m.fuse_gaussians([leaves[0]], [leaves[1]], leaves[2], leaves[3], d(g["depths"]), d(g["extrinsics"])[None],
                           d(g["intrinsics"])[None], (int(g["h"]), int(g["w"])))
    assert out[0].shape == gg["w_latent"].shape
    sum((o * d(gg[k])).sum() for o, k in zip(out, ("w_latent", "w_xyz", "w_extrinsics", "w_depths"))).backward()
    rel = lambda a, b: float((a.cpu() - b).abs().max()) / (float(b.abs().max()) + 1e-30)
    for t, key in zip(leaves, ("d_latents", "d_coords", "d_densities", "d_weights")):
        assert t.grad.shape == gg[key].shape and rel(t.grad, gg[key]) < 1e-3, (key, rel(t.grad, gg[key]))
    for k, p_ in m.gru.named_parameters():
        want = gg["d_gru__" + k.replace(".", "__")]
        assert rel(p_.grad, want) < 2e-3, (k, rel(p_.grad, want))


def test_gru_module_forward_and_backward_run_on_the_kernels(hip_device):
    """GRU.forward (networks.py:201-214's call signature, leading dims kept) runs fs_ptf_gru_forward and, under autograd,
    fs_ptf_gru_backward + fs_ptf_gru_weight_grads: output and every gradient against the oracle's GRU; a CPU call raises."""
    from freesplat_amd import ptf as P
    torch.manual_seed(5)
    gru = P.GRU().to(hip_device)
    n = 777
    x, hid = (torch.randn(1, n, 1, 64, device=hip_device, requires_grad=True) for _ in range(2))
    xe, he = (torch.sin(3 * torch.randn(1, n, 1, 24, device=hip_device)).requires_grad_(True) for _ in range(2))
    out = gru(x, hid, xe, he)
    assert out.shape == (1, n, 1, 64)
    g = torch.randn_like(out)
    got = torch.autograd.grad(out, [x, hid, xe, he] + list(gru.parameters()), g)
    ins = [t.detach().clone().reshape(n, -1).requires_grad_(True) for t in (x, hid, xe, he)]
    ps = {k: v.detach().clone().requires_grad_(True) for k, v in gru.named_parameters()}
    from oracle.ptf_oracle import gru as ogru
    ref_out = ogru(ps, *ins)
    assert (out.reshape(n, 64) - ref_out).abs().max().item() < 1e-5
    ref = torch.autograd.grad(ref_out, ins + [ps[k] for k, _ in gru.named_parameters()], g.reshape(n, 64))
    for a, b in zip(got, ref):
        assert (a.reshape(b.shape) - b).abs().max().item() <= 1e-4 * (b.abs().max().item() + 1e-20)
    with pytest.raises(RuntimeError, match="no CPU path"):
        P.GRU()(x.cpu(), hid.cpu(), xe.cpu(), he.cpu())


def test_training_fold_with_and_without_state_trims(hip_device, monkeypatch):
    """_PtfFold.forward keeps the steps' worst-case state buffers when they are small (FREESPLAT_PTF_KEEP_BYTES, round 5) and trims
    them to the rows that exist above that: both ways the outputs and every gradient are the same bits (the backward reads the
    same rows either way)."""
    from freesplat_amd import ptf as P
    V, h, w = 4, 48, 64
    E, Kn, depths, lat, dens, wts, coords = _scene(V, h, w, seed=77)
    torch.manual_seed(3)
    m = P.PixelwiseTripletFusion().to(hip_device)
    d = lambda t: t.to(hip_device)
    gen = torch.Generator().manual_seed(5)

    def run():
        ins = [d(t).requires_grad_(True) for t in (lat, coords, dens, wts, depths)]
        out = m.fuse_gaussians([ins[0]], [ins[1]], ins[2], ins[3], ins[4], d(E)[None], d(Kn)[None], (h, w))
        cot = [torch.randn(o.shape, generator=torch.Generator().manual_seed(11 + k)).to(hip_device) for k, o in enumerate(out)]
        grads = torch.autograd.grad(out, ins + list(m.gru.parameters()), cot, allow_unused=True)
        return [o.detach() for o in out], grads
    monkeypatch.setattr(P, "_KEEP_BYTES_ENV", str(1 << 40))
    out_keep, g_keep = run()
    monkeypatch.setattr(P, "_KEEP_BYTES_ENV", "0")
    out_trim, g_trim = run()
    for a, b in zip(out_keep, out_trim):
        assert torch.equal(a, b)
    for k, (a, b) in enumerate(zip(g_keep[:5], g_trim[:5])):       # (input gradients: plain stores / fixed-order sums per row)
        assert (a is None) == (b is None) and (a is None or torch.equal(a, b)), k
    for a, b in zip(g_keep[5:], g_trim[5:]):                          # (GRU weights: fixed-order partial sums -> same bits)
        assert torch.equal(a, b)


def test_training_fold_with_kept_and_with_recomputed_gru_activations(hip_device, monkeypatch):
    """Round 6: a training fold keeps the GRU's hidden activations and gates (fs_ptf_fold_step_save) and its backward runs the transposed
    layers only (fs_ptf_gru_backward_saved); FREESPLAT_GRU_SAVE=0 re-runs the forward inside the backward kernel as before.  Same
    outputs (the same forward kernel arithmetic) and the same gradients -- the kept values ARE the values the re-run produces, up to
    the order in which the two kernels round sigmoid / tanh inputs (identical operand rows, identical MFMA order: in practice bits)."""
    from freesplat_amd import _lib, ptf as P
    if _lib.lib().fs_ptf_gru_stream_t_rows() == 0:
        pytest.skip("32-pair GRU kernels selected (FS_GRU_FWD16=0 / FS_GRU_BWD16=0): no saved-activation backward")
    V, h, w = 4, 48, 64
    E, Kn, depths, lat, dens, wts, coords = _scene(V, h, w, seed=78)
    torch.manual_seed(4)
    m = P.PixelwiseTripletFusion().to(hip_device)
    d = lambda t: t.to(hip_device)

    def run():
        ins = [d(t).requires_grad_(True) for t in (lat, coords, dens, wts, depths)]
        out = m.fuse_gaussians([ins[0]], [ins[1]], ins[2], ins[3], ins[4], d(E)[None], d(Kn)[None], (h, w))
        cot = [torch.randn(o.shape, generator=torch.Generator().manual_seed(21 + k)).to(hip_device) for k, o in enumerate(out)]
        grads = torch.autograd.grad(out, ins + list(m.gru.parameters()), cot, allow_unused=True)
        return [o.detach() for o in out], grads
    monkeypatch.setenv("FREESPLAT_GRU_SAVE", "1")
    assert P.save_gru_activations()
    out_s, g_s = run()
    monkeypatch.setenv("FREESPLAT_GRU_SAVE", "0")
    assert not P.save_gru_activations()
    out_r, g_r = run()
    for a, b in zip(out_s, out_r):
        assert torch.equal(a, b)
    for k, (a, b) in enumerate(zip(g_s, g_r)):
        assert (a is None) == (b is None), k
        if a is not None:
            scale = b.abs().max().item() + 1e-30
            assert (a - b).abs().max().item() <= 1e-5 * scale, (k, (a - b).abs().max().item(), scale)
