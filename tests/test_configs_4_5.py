"""Parity cases for BASELINE.json configs 4 and 5 (the multi-view configs; sizes scaled so the oracles
finish in seconds): 10 context views with the 9 pose-nearest as cost-volume sources (K = 8) and a 10-view
PTF fold; a 30-view long-sequence fold; fp16-stored SH coefficients in the rasterizer."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
pytestmark = pytest.mark.gpu


def test_config4_cost_volume_10_views_9_nearest(hip_device):
    import inputs
    from oracle import cost_volume_oracle as cvo
    from freesplat_amd.cost_volume import AVGFeatureVolumeManager
    from freesplat_amd.encoder_glue import prepare_cost_volume_inputs
    V, h4, w4, D, C = 10, 24, 32, 16, 48
    E, Kn = inputs.cameras(V, h4, w4, baseline=1.2, seed=11)
    feats = torch.randn(V, C, h4, w4, generator=torch.Generator().manual_seed(2))
    kw = prepare_cost_volume_inputs(E[None], Kn[None], feats, torch.full((1, V), 0.5), torch.full((1, V), 15.0),
                                    (4 * h4, 4 * w4), num_context_views=9)
    assert kw["src_feats"].shape == (V, 8, C, h4, w4)
    torch.manual_seed(3)
    m = AVGFeatureVolumeManager(h4, w4, num_depth_bins=D, mlp_channels=[202, 32, 32, 1], matching_dim_size=C)
    sd = {k.replace(".", "__"): v for k, v in m.state_dict().items()}
    ref = cvo.cost_volume(kw["cur_feats"], kw["src_feats"], kw["src_extrinsics"], kw["src_Ks"], kw["cur_invK"],
                          kw["min_depth"], kw["max_depth"], D, cvo.mlp_from_state(sd))
    with torch.no_grad():
        out = m.to(hip_device)(**{k: v.to(hip_device) for k, v in kw.items()}).cpu()
    err = (out - ref).abs()
    assert int((err > 1e-4).sum()) <= 2 and float(err.median()) < 1e-5   # (validity flips at image borders aside)


@pytest.mark.parametrize("V,h,w", [(10, 48, 64), (30, 24, 32), (30, 96, 128)])
def test_config4_5_long_sequence_fold(hip_device, V, h, w):
    """Config 4's 10 views and config 5's 30-view long sequence, EVERY view compared (count, order, values): the 30-view fold
    also at 96x128 (368 640 raw Gaussians; VERDICT r4 item 8 -- the bench-size fold is compared over 10 views in
    bench_encoder.bench_ptf's parity leg)."""
    from oracle import ptf_oracle as po
    from freesplat_amd.ptf import PixelwiseTripletFusion
    from test_ptf_hip import _scene
    E, Kn, depths, lat, dens, wts, coords = _scene(V, h, w, seed=80 + V)
    torch.manual_seed(4)
    m = PixelwiseTripletFusion()
    params = {k: v.detach().clone() for k, v in m.gru.state_dict().items()}
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    from freesplat_amd.ptf import world_to_camera
    w2c = world_to_camera(E.to(hip_device)).view(-1, 4, 4).cpu() if h * w > 4096 else None   # (see the full-size test below)
    with torch.no_grad():
        ref = po.fuse_gaussians(params, lat, coords, dens, wts, depths, E[None], Kn[None], (h, w), w2c_all=w2c)
    m = m.to(hip_device)
    d = lambda t: t.to(hip_device)
    with torch.no_grad():
        out = m.fuse_gaussians([d(lat)], [d(coords)], d(dens), d(wts), d(depths), d(E)[None], d(Kn)[None], (h, w))
    assert out[0].shape == ref[0].shape and out[0].shape[1] < V * h * w // 2   # sub-linear growth
    for a, b, name in zip(out, ref, ("latent", "xyz", "extrinsics", "depths")):
        assert (a.cpu() - b.detach()).abs().max().item() <= 1e-4, name


@pytest.mark.slow
def test_config4_cost_volume_at_its_real_size(hip_device):
    """BASELINE config 4 at the size bench.py times it (`fvt10_96x128_K8`: 10 context views at the native 96x128
    matching resolution, D = 128 planes, the 9 pose-nearest views as sources -> K = 8): the whole HIP call, ONE of its
    current views checked against the reference-pinned oracle (a 256-thread host needs ~10 s per view)."""
    import inputs
    from oracle import cost_volume_oracle as cvo
    from freesplat_amd.cost_volume import AVGFeatureVolumeManager
    V, K, h4, w4, D, C = 10, 8, 96, 128, 128, 48
    kw = inputs.cv_inputs(V, K, h4, w4, C, seed=1)             # (bench_encoder.bench_cost_volume's inputs)
    torch.manual_seed(0)
    m = AVGFeatureVolumeManager(h4, w4, num_depth_bins=D, mlp_channels=[202, 32, 32, 1], matching_dim_size=C)
    sd = {k.replace(".", "__"): v for k, v in m.state_dict().items()}
    with torch.no_grad():
        out = m.to(hip_device)(**{k: v.to(hip_device) for k, v in kw.items()}).cpu()
    assert out.shape == (V, D, h4, w4)
    torch.set_num_threads(os.cpu_count() or 1)
    for v in (3,):
        one = slice(v, v + 1)
        ref = cvo.cost_volume(kw["cur_feats"][one], kw["src_feats"][one], kw["src_extrinsics"][one], kw["src_Ks"][one],
                              kw["cur_invK"][one], kw["min_depth"], kw["max_depth"], D, cvo.mlp_from_state(sd))
        err = (out[one] - ref).abs()
        # cells above 1e-4 are validity flips of bilinear taps on the image border (tests/test_cost_volume_hip.py)
        assert int((err > 1e-4).sum()) <= max(2, err.numel() // 100000) and float(err.median()) < 1e-5, (
            int((err > 1e-4).sum()), float(err.max()), float(err.median()))


@pytest.mark.slow
@pytest.mark.parametrize("V", [5, 10])
def test_config4_fold_at_its_real_size(hip_device, V):
    """BASELINE config 4's fold at the size bench.py times it (`fold_10_views`: 10 views at 384x512 = 1.97 M raw
    Gaussians): same count, same ORDER (the appended / kept / fused layout of every step) and values within 1e-4 of the
    reference-pinned oracle.  (Round 3 skipped the 10-view case by default: its oracle fold took 2.5 - 5 minutes on the
    GPU box -- with torch set to all 256 host threads, whose synchronisation dominates the fold's many small operations;
    on 16 threads it takes seconds, so both cases run in the default suite.)"""
    from oracle import ptf_oracle as po
    from freesplat_amd.ptf import PixelwiseTripletFusion
    from test_ptf_hip import _scene
    h, w = 384, 512
    E, Kn, depths, lat, dens, wts, coords = _scene(V, h, w, seed=5)          # (bench_encoder.bench_ptf's scene)
    torch.manual_seed(1)
    m = PixelwiseTripletFusion()
    params = {k: v.detach().clone() for k, v in m.gru.state_dict().items()}
    mg = PixelwiseTripletFusion()
    mg.load_state_dict(m.state_dict())
    mg = mg.to(hip_device)
    d = lambda t: t.to(hip_device)
    with torch.no_grad():
        out = [x.cpu() for x in mg.fuse_gaussians([d(lat)], [d(coords)], d(dens), d(wts), d(depths), d(E)[None], d(Kn)[None], (h, w))]
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    # Both sides round their pixels from the SAME world-to-camera matrices (the product's: torch's batched inverse on the
    # GPU): at ~10^6 projections per step one of them lands within an ulp of a rounding boundary, and the host LAPACK's
    # inverse differs from the GPU solver's in the last bit (5 views of this scene: 325 619 vs 325 618 Gaussians).  What
    # is under test is the fold, not the two LU implementations.
    from freesplat_amd.ptf import world_to_camera
    w2c = world_to_camera(E.to(hip_device)).view(-1, 4, 4).cpu()
    with torch.no_grad():
        ref = po.fuse_gaussians(params, lat, coords, dens, wts, depths, E[None], Kn[None], (h, w), w2c_all=w2c)
    assert out[0].shape == ref[0].shape and out[0].shape[1] < V * h * w        # same count; something fused
    for a, b, name in zip(out, ref, ("latent", "xyz", "extrinsics", "depths")):
        assert a.shape == b.shape, name
        assert (a - b).abs().max().item() <= 1e-4, name          # row i of ours is row i of the reference: same order


@pytest.mark.slow
def test_config4_fold_real_size_oracle_inverts_for_itself(hip_device):
    """The same full-size fold (5 views at 384x512) END TO END INDEPENDENT: the oracle inverts the extrinsics itself (host
    LAPACK), the product on the GPU (fs_invert_4x4) -- nothing is shared between the two sides.  The two
    inverses differ in the last bit, and of ~10^6 projections per step one may land on the other side of a rounding boundary
    (325 619 vs 325 618 Gaussians on this scene in round 3), so rows cannot be compared by position: the counts may differ by
    <= 2, and every row is matched by its position in space -- all but a handful (the flipped pixel's Gaussians and their fusion
    partners: <= 64 of 1.3 M) must exist on both sides and agree to 1e-4 in every output."""
    from scipy.spatial import cKDTree
    from oracle import ptf_oracle as po
    from freesplat_amd.ptf import PixelwiseTripletFusion
    from test_ptf_hip import _scene
    V, h, w = 5, 384, 512
    E, Kn, depths, lat, dens, wts, coords = _scene(V, h, w, seed=5)
    torch.manual_seed(1)
    m = PixelwiseTripletFusion()
    params = {k: v.detach().clone() for k, v in m.gru.state_dict().items()}
    mg = PixelwiseTripletFusion()
    mg.load_state_dict(m.state_dict())
    mg = mg.to(hip_device)
    d = lambda t: t.to(hip_device)
    with torch.no_grad():
        out = [x.cpu() for x in mg.fuse_gaussians([d(lat)], [d(coords)], d(dens), d(wts), d(depths), d(E)[None], d(Kn)[None], (h, w))]
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        ref = po.fuse_gaussians(params, lat, coords, dens, wts, depths, E[None], Kn[None], (h, w))      # (its own inverse)
    n_out, n_ref = out[0].shape[1], ref[0].shape[1]
    assert abs(n_out - n_ref) <= 2 and n_out < V * h * w, (n_out, n_ref)
    xyz_o, xyz_r = out[1][0].numpy().astype(np.float64), ref[1][0].detach().numpy().astype(np.float64)
    dist, idx = cKDTree(xyz_r).query(xyz_o)
    matched = dist <= 1e-4
    # (measured on the MI355X box: 24 of 1.3 M rows without a partner -- one flipped pixel changes which Gaussians fuse, and every
    #  later step that touches the fused row inherits the difference)
    assert int((~matched).sum()) <= 64, int((~matched).sum())
    lat_o, lat_r = out[0][0].numpy(), ref[0][0].detach().numpy()
    dep_o, dep_r = out[3].reshape(-1).numpy(), ref[3].detach().reshape(-1).numpy()
    lat_err = np.abs(lat_o[matched] - lat_r[idx[matched]]).max(axis=1)
    dep_err = np.abs(dep_o[matched] - dep_r[idx[matched]]) if dep_o.shape[0] == n_out and dep_r.shape[0] == n_ref else np.zeros(1)
    # (two Gaussians closer than 1e-4 in space could be matched crosswise: allowed for, never seen)
    assert int((lat_err > 1e-4).sum()) <= 64 and int((dep_err > 1e-4).sum()) <= 64, (float(lat_err.max()), float(dep_err.max()))
    # and the other direction: every reference row exists in ours
    dist_r, _ = cKDTree(xyz_o).query(xyz_r)
    assert int((dist_r > 1e-4).sum()) <= 64


def test_config5_fp16_sh_storage(hip_device):
    """fp16 SH = storage only (BASELINE config 5): against the ORACLE fed the fp16-rounded coefficients the image is
    bit-exact and the gradients are within the backward's bar; and the fp32 HIP path on those coefficients gives
    the same image bit for bit."""
    from oracle import raster_oracle as ro
    from util_raster import hip_forward, oracle_forward, small_scene, view_inputs
    H, W = 64, 80
    scene, cams = small_scene(N=1500, H=H, W=W, seed=12)
    vi = view_inputs(scene, cams, 0, H, W, bg=(0.1, 0.2, 0.3))
    vi32 = dict(vi); vi32["shs"] = vi["shs"].half().float()
    vi16 = dict(vi); vi16["shs"] = vi["shs"].half()
    assert not torch.equal(vi32["shs"], vi["shs"])                     # the rounding is not a no-op on this scene
    st = oracle_forward(vi32)
    (c32, _, d32, _), l32 = hip_forward(vi32, hip_device, requires_grad=True)
    (c16, _, d16, _), l16 = hip_forward(vi16, hip_device, requires_grad=True)
    np.testing.assert_array_equal(c16.detach().cpu().numpy(), st["color"])
    np.testing.assert_array_equal(d16.detach().cpu().numpy(), st["depth"])
    assert torch.equal(c32, c16) and torch.equal(d32, d16)
    w = torch.randn_like(c32)
    ref = ro.backward(st, w.cpu().numpy())
    (c32 * w).sum().backward()
    (c16 * w).sum().backward()
    assert l16["shs"].grad.dtype == torch.float16
    for k in ("means3D", "cov3D", "opacities"):
        s = l32[k].grad.abs().max().item()
        assert (l32[k].grad - l16[k].grad).abs().max().item() <= 2e-4 * s, k
    s = l32["shs"].grad.abs().max().item()
    assert (l32["shs"].grad - l16["shs"].grad.float()).abs().max().item() <= 2e-3 * s   # fp16 rounding of the grad
    for k in ("means3D", "cov3D", "opacities"):
        r = ref[k]
        assert np.abs(l16[k].grad.cpu().numpy().reshape(r.shape) - r).max() <= 2e-4 * np.abs(r).max(), k
    r = ref["shs"]
    assert np.abs(l16["shs"].grad.float().cpu().numpy() - r).max() <= 2e-3 * np.abs(r).max()
