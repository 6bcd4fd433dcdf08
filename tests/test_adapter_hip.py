"""GPU parity of the adapter kernels (C ABI fs_unproject_*, fs_gaussian_head_*) against the
reference's golden outputs and the oracle's autograd."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.gpu


def _load(name):
    z = np.load(os.path.join(HERE, "golden", name))
    return {k: torch.from_numpy(z[k]) if z[k].ndim else z[k].item() for k in z.files}


def _adapter(dev):
    from freesplat_amd.gaussian_adapter import GaussianAdapter, GaussianAdapterCfg
    return GaussianAdapter(GaussianAdapterCfg(0.5, 15.0, 2)).to(dev)


def test_unproject_matches_reference_and_grad(hip_device):
    from oracle import adapter_oracle as ao
    g = _load("ptf_small.npz")
    h, w = int(g["h"]), int(g["w"])
    V = g["depths"].shape[0]
    d = lambda t: t.to(hip_device)
    dep = d(g["depths"].reshape(1, V, h * w, 1, 1)).requires_grad_(True)
    ad = _adapter(hip_device)
    xyz = ad(d(g["extrinsics"])[None, :, None, None, None], d(g["intrinsics"])[None, :, None, None, None], None, dep,
             None, None, (h, w), fusion=True)
    assert xyz.shape == g["coords"].shape
    assert (xyz.detach().cpu() - g["coords"]).abs().max().item() <= 2e-6
    wgt = torch.randn(g["coords"].shape, generator=torch.Generator().manual_seed(0))
    (xyz * d(wgt)).sum().backward()
    K0 = g["intrinsics"][0].clone(); K0[0] *= w; K0[1] *= h
    dc = g["depths"].reshape(V, -1).clone().requires_grad_(True)
    ref = ao.unproject(dc, g["extrinsics"], torch.stack([K0[0, 0], K0[1, 1], K0[0, 2], K0[1, 2]]), h, w)
    (ref * wgt[0, :, :, 0, 0]).sum().backward()
    assert (dep.grad.cpu().reshape(V, -1) - dc.grad).abs().max().item() <= 1e-5 * dc.grad.abs().max().item()


def test_gaussian_head_matches_reference_golden(hip_device):
    g = _load("adapter_small.npz")
    h, w = int(g["h"]), int(g["w"])
    M = g["extrinsics"].shape[0]
    d = lambda t: t.to(hip_device)
    ad = _adapter(hip_device)
    out = ad(d(g["extrinsics"]).view(1, 1, M, 1, 1, 4, 4), d(g["intrinsics"]).view(1, 1, 1, 1, 1, 3, 3).expand(1, 1, M, 1, 1, 3, 3),
             None, d(g["depths"]).view(1, 1, M, 1, 1), d(g["opacities"]).view(1, 1, M, 1, 1), d(g["raw"]).view(1, 1, M, 1, 1, 34),
             (h, w), fusion=False, coords=d(g["coords"]).view(1, 1, M, 1, 1, 3))
    for got, key, rtol in ((out.covariances, "out_cov", 2e-5), (out.harmonics, "out_harmonics", 1e-6),
                           (out.scales, "out_scales", 2e-6), (out.rotations, "out_rotations", 2e-6),
                           (out.means, "out_means", 0), (out.opacities, "out_opacities", 0)):
        want = g[key]
        assert got.shape == want.shape, key
        assert (got.cpu() - want).abs().max().item() <= rtol * want.abs().max().item() + 1e-12, key


def test_gaussian_head_backward_vs_oracle_autograd(hip_device):
    from oracle import adapter_oracle as ao
    from freesplat_amd.gaussian_adapter import _Head
    gen = torch.Generator().manual_seed(3)
    M = 2000
    raw = torch.randn(M, 34, generator=gen)
    dep = 1.0 + torch.rand(M, generator=gen)
    E = torch.eye(4).repeat(M, 1, 1) + 0.1 * torch.randn(M, 4, 4, generator=gen)
    mult = torch.tensor([0.0123])
    mask = torch.tensor([1.0, .025, .025, .025, .00625, .00625, .00625, .00625, .00625])
    gcov, gsh = torch.randn(M, 3, 3, generator=gen), torch.randn(M, 3, 9, generator=gen)
    gsc, grot = torch.randn(M, 3, generator=gen), torch.randn(M, 4, generator=gen)
    leaf = lambda t: t.double().clone().requires_grad_(True)
    r64, d64, e64 = leaf(raw), leaf(dep), leaf(E)
    cov, sh, sc, rot = ao.gaussian_head(r64, d64, e64, mult.double()[0], mask.double())
    ((cov * gcov).sum() + (sh * gsh).sum() + (sc * gsc).sum() + (rot * grot).sum()).backward()
    dv = lambda t: t.to(hip_device)
    rg, dg, eg = dv(raw).requires_grad_(True), dv(dep).requires_grad_(True), dv(E).requires_grad_(True)
    o = _Head.apply(rg, dg, eg, dv(mult), dv(mask), 0.5, 15.0)
    for a, b in zip(o, (cov, sh, sc, rot)):
        assert (a.detach().cpu().double() - b.detach()).abs().max().item() <= 2e-5 * (b.abs().max().item() + 1e-12)
    ((o[0] * dv(gcov)).sum() + (o[1] * dv(gsh)).sum() + (o[2] * dv(gsc)).sum() + (o[3] * dv(grot)).sum()).backward()
    for got, want, name in ((rg.grad, r64.grad, "raw"), (dg.grad, d64.grad, "depths"), (eg.grad, e64.grad, "extrinsics")):
        s = want.abs().max().item()
        assert (got.cpu().double() - want).abs().max().item() <= 2e-4 * s, name


@pytest.mark.parametrize("N,h,w", [(1, 8, 8), (3, 37, 41), (2, 64, 96)])
def test_latents_pack_is_the_reference_expression(hip_device, N, h, w):
    """fs_latents_pack_forward / _backward against the glue they replace, op for op (encoder_freesplat.py:311-316 on the CPU:
    `head[:, 1:] + skip` rearranged "(b v) c h w -> b v (h w) c", densities from head[:, :1]): ONE fp32 add per element and
    pure data movement, so values AND gradients are bit-identical; tiles that end inside the image (h*w not a multiple of 64),
    a single partial tile, and gradients on one output only."""
    from freesplat_amd.gaussian_adapter import latents_pack
    g = torch.Generator().manual_seed(N * 1000 + h)
    head, skip = torch.randn(N, 65, h, w, generator=g), torch.randn(N, 64, h, w, generator=g)
    g_lat, g_dens = torch.randn(N, h * w, 64, generator=g), torch.randn(N, h * w, generator=g)

    def ref(head, skip):
        lat = (head[:, 1:] + skip).reshape(N, 64, h * w).transpose(-1, -2)
        return lat, head[:, 0].reshape(N, h * w)
    hc, sc = head.clone().requires_grad_(True), skip.clone().requires_grad_(True)
    rl, rd = ref(hc, sc)
    ((rl * g_lat).sum() + (rd * g_dens).sum()).backward()
    hd, sd = head.to(hip_device).requires_grad_(True), skip.to(hip_device).requires_grad_(True)
    lat, dens = latents_pack(hd, sd)
    assert lat.shape == (N, h * w, 64) and lat.is_contiguous() and dens.shape == (N, h * w)
    assert torch.equal(lat.detach().cpu(), rl.detach()) and torch.equal(dens.detach().cpu(), rd.detach())
    ((lat * g_lat.to(hip_device)).sum() + (dens * g_dens.to(hip_device)).sum()).backward()
    assert torch.equal(hd.grad.cpu(), hc.grad) and torch.equal(sd.grad.cpu(), sc.grad)
    # only the latents carry a gradient: the density channel of g_head must be written as zero (not left uninitialised)
    hd2 = head.to(hip_device).requires_grad_(True)
    lat2, _ = latents_pack(hd2, skip.to(hip_device))
    (lat2 * g_lat.to(hip_device)).sum().backward()
    assert bool((hd2.grad[:, 0] == 0).all()) and torch.equal(hd2.grad[:, 1:].cpu(), hc.grad[:, 1:])
    with pytest.raises(RuntimeError, match="no CPU path"):
        latents_pack(head, skip)
