#!/usr/bin/env python
"""Generates tests/golden/*.npz by IMPORTING the reference (read-only at /root/reference) in the
build container and running its own code on seeded synthetic inputs.  Run here only:

    python tests/golden/make_golden.py

The reference's Python never travels to the GPU box; only the arrays written here do.  Nothing
below copies reference source: it stubs the reference's missing third-party imports
(SURVEY.md Appendix D), imports its modules, and calls them.

Fixtures (fp32, torch.manual_seed, sizes per SURVEY.md 8(c)):
  cv_small_k1.npz / cv_small_k2.npz / cv_small_c16.npz   AVGFeatureVolumeManager.forward (cost_volume.py:351-381, 429-619)
  cv_native_stat.json                  96x128, D=128 summary statistics + SHA-256 of the output
  ptf_small.npz / ptf_tie.npz          EncoderFreeSplat.fuse_gaussians (encoder_freesplat.py:431-522)
  adapter_small.npz                    GaussianAdapter.forward fusion=True / False (gaussian_adapter.py:135-201)
  depth_tail_{log,inv}.npz             DepthDecoder tail (networks.py:130-152) on the real module's logits
  glue_small.npz                       calculate_distance_matrix (encoder_freesplat.py:50-60)
  framing.npz                          get_fov / get_projection_matrix + render_cuda's matrices
                                       (projection.py:233-247, cuda_splatting.py:17-87)
"""
import hashlib
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
# python make_golden.py [fixture names ...]: regenerate only those cost-volume fixtures (the others stay byte-identical)
ONLY = set(sys.argv[1:])
sys.path.insert(0, OUT)
from inputs import cameras, cv_inputs, ptf_inputs  # noqa: E402  (seeded input generators shared with the tests)


def install_shim():
    stub_dir = tempfile.mkdtemp(prefix="fs_stub_")
    os.makedirs(os.path.join(stub_dir, "kornia"))
    with open(os.path.join(stub_dir, "kornia", "__init__.py"), "w") as f:
        f.write("from . import filters\n")
    with open(os.path.join(stub_dir, "kornia", "filters.py"), "w") as f:
        f.write("import torch\n"
                "def blur_pool2d(x: torch.Tensor, kernel_size: int) -> torch.Tensor:\n    return x\n"
                "def gaussian_blur2d(*a, **k):\n    raise NotImplementedError\n"
                "def spatial_gradient(*a, **k):\n    raise NotImplementedError\n")
    sys.path[:0] = [stub_dir, REF]

    class _Sub:
        def __class_getitem__(cls, item):
            return item[0] if isinstance(item, tuple) else item

    jt = types.ModuleType("jaxtyping")
    for n in ("Float", "Int64", "Bool", "Int", "UInt8", "Shaped"):
        setattr(jt, n, _Sub)
    jt.install_import_hook = None
    sys.modules["jaxtyping"] = jt
    for name in ("torchvision", "torchvision.models", "torchvision.transforms",
                 "torchvision.transforms.functional", "timm", "cv2", "e3nn", "e3nn.o3"):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    sys.modules["e3nn.o3"].matrix_to_angles = None
    sys.modules["e3nn.o3"].wigner_D = None
    sys.modules["e3nn"].o3 = sys.modules["e3nn.o3"]

    def pkg(name, path):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, path)]
        sys.modules[name] = m
        return m

    pkg("src", "src")
    pkg("src.model", "src/model")
    pkg("src.model.encoder", "src/model/encoder")
    b = pkg("src.model.encoder.backbone", "src/model/encoder/backbone")
    b.Backbone = b.BackboneCfg = object
    b.get_backbone = None
    pkg("src.model.encoder.visualization", "src/model/encoder/visualization")
    pkg("src.model.encoder.epipolar", "src/model/encoder/epipolar")
    pkg("src.model.decoder", "src/model/decoder")
    pkg("src.dataset", "src/dataset")
    pkg("src.dataset.shims", "src/dataset/shims")
    m = types.ModuleType("src.dataset.shims.patch_shim"); m.apply_patch_shim = None
    sys.modules[m.__name__] = m
    m = types.ModuleType("src.dataset.types"); m.BatchedExample = dict; m.DataShim = object; m.BatchedViews = dict
    sys.modules[m.__name__] = m
    m = types.ModuleType("src.model.encoder.visualization.encoder_visualizer_epipolar_cfg")
    m.EncoderVisualizerEpipolarCfg = object
    sys.modules[m.__name__] = m
    m = types.ModuleType("diff_gaussian_rasterization_depth")
    m.GaussianRasterizationSettings = None; m.GaussianRasterizer = None
    sys.modules[m.__name__] = m


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name), **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                                     for k, v in arrs.items()})
    print("wrote", name, {k: tuple(np.asarray(v.detach().cpu() if torch.is_tensor(v) else v).shape) for k, v in arrs.items()})


def gen_cost_volume():
    from src.model.encoder.modules.cost_volume import AVGFeatureVolumeManager
    for name, V, K, behind, C in (("cv_small_k1", 2, 1, False, 48), ("cv_small_k2", 3, 2, True, 48),
                                  ("cv_small_c16", 3, 2, True, 16)):   # 16: the module's default matching dimension
        if ONLY and name not in ONLY:
            continue
        h4, w4, D = 12, 16, 8
        torch.manual_seed(100 + V)
        cv = AVGFeatureVolumeManager(matching_height=h4, matching_width=w4, num_depth_bins=D,
                                     mlp_channels=[202, 32, 32, 1], matching_dim_size=C).eval()
        kw = cv_inputs(V, K, h4, w4, C, seed=200 + V, behind=behind)
        with torch.no_grad():
            out = cv(**kw)
        sd = {k.replace(".", "__"): v for k, v in cv.state_dict().items()}
        save(name + ".npz", out=out, D=D, **kw, **sd)
    if ONLY:
        return
    # native-size statistics
    h4, w4, D, C, V, K = 96, 128, 128, 48, 2, 1
    torch.manual_seed(7)
    cv = AVGFeatureVolumeManager(matching_height=h4, matching_width=w4, num_depth_bins=D,
                                 mlp_channels=[202, 32, 32, 1], matching_dim_size=C).eval()
    kw = cv_inputs(V, K, h4, w4, C, seed=8)
    with torch.no_grad():
        out = cv(**kw)
    o = out.numpy()
    stat = dict(shape=list(o.shape), mean=float(o.mean()), abs_mean=float(np.abs(o).mean()), std=float(o.std()),
                min=float(o.min()), max=float(o.max()), sha256=hashlib.sha256(o.tobytes()).hexdigest(),
                seed_module=7, seed_inputs=8, probe=[float(x) for x in o[0, ::16, 40, 60]])
    json.dump(stat, open(os.path.join(OUT, "cv_native_stat.json"), "w"), indent=1)
    print("wrote cv_native_stat.json", stat["shape"], stat["mean"])


def gen_ptf_and_adapter():
    from src.model.encoder.encoder_freesplat import EncoderFreeSplat
    from src.model.encoder.modules.networks import GRU
    from src.model.encoder.common.gaussian_adapter import GaussianAdapter, GaussianAdapterCfg
    from src.geometry.projection import sample_image_grid
    for name, V, tie in (("ptf_small", 3, False), ("ptf_tie", 2, True)):
        h, w = 24, 32
        E, Kn, depths, lat, dens, wts = ptf_inputs(V, h, w, seed=300 + V, tie=tie)
        torch.manual_seed(400 + V)
        gru = GRU().eval()
        adapter = GaussianAdapter(GaussianAdapterCfg(0.5, 15.0, 2))
        with torch.no_grad():
            # argument shapes exactly as encoder_freesplat.py:317-326 passes them
            xy_ray, _ = sample_image_grid((h, w), torch.device("cpu"))
            xy_ray = xy_ray.reshape(h * w, 1, 2)[None, None].expand(1, V, h * w, 1, 2)
            coords = adapter.forward(E[None, :, None, None, None], Kn[None, :, None, None, None],
                                     xy_ray[:, :, :, :, None, :], depths.view(1, V, h * w, 1, 1), dens, lat, (h, w),
                                     fusion=True)
            out = EncoderFreeSplat.fuse_gaussians(types.SimpleNamespace(gru=gru), [lat], [coords], dens, wts,
                                                  depths.view(V, 1, h, w), E[None], Kn[None], (h, w))
        sd = {"gru__" + k.replace(".", "__"): v for k, v in gru.state_dict().items()}
        save(name + ".npz", latents=lat, coords=coords, densities=dens, weights=wts, depths=depths,
             extrinsics=E, intrinsics=Kn, h=h, w=w, out_latent=out[0], out_xyz=out[1], out_extrinsics=out[2],
             out_depths=out[3], **sd)
    # adapter, fusion=False
    h, w, V = 8, 12, 2
    torch.manual_seed(500)
    adapter = GaussianAdapter(GaussianAdapterCfg(0.5, 15.0, 2))
    M = 40
    E = torch.eye(4).repeat(M, 1, 1) + 0.05 * torch.randn(M, 4, 4)  # blended, non-rigid "extrinsics"
    Kn = torch.tensor([[0.9, 0, 0.49], [0, 1.2, 0.51], [0, 0, 1]])
    raw = torch.randn(1, M, 1, 7 + 27)
    dep = 1.0 + torch.rand(1, M)
    opa = torch.rand(1, M, 1, 1)
    xyz = torch.randn(1, M, 3)
    with torch.no_grad():
        # argument shapes exactly as encoder_freesplat.py:376-386 passes them
        g = adapter.forward(E.view(1, 1, M, 1, 1, 4, 4), Kn.view(1, 1, 1, 1, 1, 3, 3).expand(1, 1, M, 1, 1, 3, 3), None,
                            dep.view(1, 1, M, 1, 1), opa.view(1, 1, M, 1, 1), raw.view(1, 1, M, 1, 1, 34), (h, w),
                            fusion=False, coords=xyz.view(1, 1, M, 1, 1, 3))
    save("adapter_small.npz", extrinsics=E, intrinsics=Kn, raw=raw, depths=dep, opacities=opa, coords=xyz, h=h, w=w,
         out_means=g.means, out_cov=g.covariances, out_harmonics=g.harmonics, out_opacities=g.opacities,
         out_scales=g.scales, out_rotations=g.rotations, sh_mask=adapter.sh_mask)


def gen_framing():
    from src.model.decoder.cuda_splatting import get_projection_matrix
    from src.geometry.projection import get_fov
    V = 3
    E, Kn = cameras(V, 10, 10, seed=9)
    E[:, :3, 3] += torch.tensor([0.1, -0.2, 0.3])
    near, far = torch.tensor([0.5, 0.5, 0.25]), torch.tensor([15.0, 15.0, 10.0])
    fov = get_fov(Kn)
    scale = 1 / near
    Es = E.clone()
    Es[..., :3, 3] = Es[..., :3, 3] * scale[:, None]
    P = get_projection_matrix(near * scale, far * scale, fov[:, 0], fov[:, 1])
    view = Es.inverse().transpose(1, 2)
    full = view @ P.transpose(1, 2)
    save("framing.npz", extrinsics=E, intrinsics=Kn, near=near, far=far, fov=fov, projection=P, view=view, full=full,
         tan=(0.5 * fov).tan(), campos=Es[:, :3, 3])


def gen_glue():
    """calculate_distance_matrix (encoder_freesplat.py:50-60) on a 6-view pose set."""
    from src.model.encoder.encoder_freesplat import calculate_distance_matrix
    E, _ = cameras(6, 10, 10, baseline=1.5, seed=4)
    E[3, :3, 3] += torch.tensor([0.4, 0.0, 0.2])
    d = calculate_distance_matrix(E[None])
    save("glue_small.npz", extrinsics=E, dist=d)


def gen_depth_tail():
    """The real DepthDecoder (networks.py:19-154) on random feature pyramids; the fixture keeps the finest
    scale's plane logits (conv_depth['0'] of output_pred_s0) and the tail's outputs (:130-152)."""
    from src.model.encoder.modules.networks import DepthDecoder
    for name, log_planes in (("depth_tail_log.npz", True), ("depth_tail_inv.npz", False)):
        torch.manual_seed(600 + int(log_planes))
        dd = DepthDecoder([24, 64, 128, 256, 384], num_output_channels=1 + 64, near=0.5, far=15.0, num_samples=32,
                          log_planes=log_planes).eval()
        feats = [torch.randn(2, c, 32 >> i, 48 >> i) for i, c in enumerate([24, 64, 128, 256, 384])]
        with torch.no_grad():
            out = dd(feats)
            logits = dd.conv_depth["0"](out["output_pred_s0_b1hw"]) * 4.0   # (scaled: sharper softmax)
            # re-run the tail on the scaled logits with the reference's own ops (lines 131-152)
            import torch.nn.functional as F
            planes = F.softmax(logits, dim=1)
            coarse = (dd.depth_candi_curr * planes).sum(dim=1, keepdim=True)
            fine = F.interpolate(coarse, scale_factor=2, mode="bilinear", align_corners=True)
            wts = F.interpolate(planes, scale_factor=2, mode="bilinear", align_corners=True).max(dim=1, keepdim=True)[0]
            depth = torch.exp(coarse) if log_planes else 1.0 / coarse
            dmap = torch.exp(fine) if log_planes else 1.0 / fine
            # and check that un-scaled logits reproduce the module's own outputs exactly
            p0 = F.softmax(dd.conv_depth["0"](out["output_pred_s0_b1hw"]), dim=1)
            c0 = (dd.depth_candi_curr * p0).sum(dim=1, keepdim=True)
            assert torch.equal(c0, out["log_depth_pred_s0_b1hw"])
        save(name, logits=logits, candidates=dd.depth_candi_curr.reshape(-1), log_planes=int(log_planes), coarse=coarse,
             depth=depth, depth_map=dmap, depth_weights=wts, module_logits=dd.conv_depth["0"](out["output_pred_s0_b1hw"]).detach(),
             module_log_depth=out["log_depth_pred_s0_b1hw"], module_depth_map=out["depth_pred_s-1_b1hw"],
             module_depth_weights=out["depth_weights"])


def gen_ply():
    """The reference's export_ply (src/model/ply_export.py:26-92) run for real on a seeded scene.  `plyfile` is not
    installed here, so a recording stand-in for its two entry points captures what the reference hands to
    PlyElement.describe (the structured vertex array = its attribute table and names) and the output path; the
    fixture keeps the inputs and that table.  (The byte layout plyfile would write -- binary_little_endian, one
    `property float <name>` per field in dtype order -- is the public PLY format, restated in freesplat_amd.ply_export.)"""
    rec = {}
    m = types.ModuleType("plyfile")

    class PlyElement:
        @staticmethod
        def describe(elements, name):
            rec["elements"], rec["name"] = elements.copy(), name
            return ("element", name)

    class PlyData:
        def __init__(self, elements):
            rec["n_elements"] = len(elements)

        def write(self, path):
            rec["path"] = str(path)
    m.PlyElement, m.PlyData = PlyElement, PlyData
    sys.modules["plyfile"] = m
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_ply_export", os.path.join(REF, "src", "model", "ply_export.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    from pathlib import Path
    g = torch.Generator().manual_seed(11)
    G = 400
    a, b = 0.4, -0.2
    Ry = torch.tensor([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], dtype=torch.float32)
    Rx = torch.tensor([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]], dtype=torch.float32)
    E = torch.eye(4); E[:3, :3] = Ry @ Rx; E[:3, 3] = torch.tensor([0.3, -0.1, 1.2])
    means = torch.randn(G, 3, generator=g) * torch.tensor([2.0, 1.0, 3.0]) + torch.tensor([0.5, -1.0, 4.0])
    scales = torch.rand(G, 3, generator=g) * 0.05 + 0.001
    q = torch.randn(G, 4, generator=g); q = q / q.norm(dim=-1, keepdim=True)
    sh = torch.randn(G, 3, 9, generator=g)
    op = torch.rand(G, generator=g)
    out_dir = tempfile.mkdtemp(prefix="fs_ply_")
    mod.export_ply(E, means, scales, q, sh, op, Path(out_dir) / "sub" / "scene.ply")
    el = rec["elements"]
    assert rec["name"] == "vertex" and rec["n_elements"] == 1 and rec["path"].endswith("scene.ply")
    table = np.stack([el[n] for n in el.dtype.names], axis=1).astype(np.float32)
    assert all(el.dtype[n] == np.dtype("f4") for n in el.dtype.names)
    save("ply_small", extrinsics=E, means=means, scales=scales, rotations=q, harmonics=sh, opacities=op,
         table=table, names=np.array(el.dtype.names))


def gen_backward():
    """Gradients computed by autograd THROUGH THE REFERENCE'S OWN MODULES (round 4): the cost volume's w.r.t. both feature
    maps and the six MLP tensors for a seeded grad_out, and the fold's w.r.t. latents, coordinates, densities, weights and
    the 12 GRU tensors for seeded output weights -- they pin the oracles' autograd (CPU) and the HIP backward kernels (GPU)
    to the reference directly, not only through the equality of the forwards."""
    from src.model.encoder.modules.cost_volume import AVGFeatureVolumeManager
    from src.model.encoder.encoder_freesplat import EncoderFreeSplat
    from src.model.encoder.modules.networks import GRU
    from src.model.encoder.common.gaussian_adapter import GaussianAdapter, GaussianAdapterCfg
    from src.geometry.projection import sample_image_grid
    # ---- cost volume: the cv_small_k2 case (3 views, 2 sources, one of them turned round) ----
    V, K, C, h4, w4, D = 3, 2, 48, 12, 16, 8
    torch.manual_seed(100 + V)
    cv = AVGFeatureVolumeManager(matching_height=h4, matching_width=w4, num_depth_bins=D, mlp_channels=[202, 32, 32, 1],
                                 matching_dim_size=C).eval()
    kw = cv_inputs(V, K, h4, w4, C, seed=200 + V, behind=True)
    kw["cur_feats"].requires_grad_(True)
    kw["src_feats"].requires_grad_(True)
    out = cv(**kw)
    g = torch.randn(out.shape, generator=torch.Generator().manual_seed(77))
    (out * g).sum().backward()
    net = cv.mlp.net
    save("cv_small_k2_grads.npz", grad_out=g, out=out, d_cur_feats=kw["cur_feats"].grad, d_src_feats=kw["src_feats"].grad,
         d_w1=net[0].weight.grad, d_b1=net[0].bias.grad, d_w2=net[2].weight.grad, d_b2=net[2].bias.grad,
         d_w3=net[4].weight.grad, d_b3=net[4].bias.grad)
    # ---- PTF: the ptf_small case (3 views, 24 x 32) ----
    V, h, w = 3, 24, 32
    E, Kn, depths, lat, dens, wts = ptf_inputs(V, h, w, seed=300 + V, tie=False)
    torch.manual_seed(400 + V)
    gru = GRU().eval()
    adapter = GaussianAdapter(GaussianAdapterCfg(0.5, 15.0, 2))
    with torch.no_grad():
        xy_ray, _ = sample_image_grid((h, w), torch.device("cpu"))
        xy_ray = xy_ray.reshape(h * w, 1, 2)[None, None].expand(1, V, h * w, 1, 2)
        coords = adapter.forward(E[None, :, None, None, None], Kn[None, :, None, None, None], xy_ray[:, :, :, :, None, :],
                                 depths.view(1, V, h * w, 1, 1), dens, lat, (h, w), fusion=True)
    leaves = [t.clone().requires_grad_(True) for t in (lat, coords, dens, wts)]
    out = EncoderFreeSplat.fuse_gaussians(types.SimpleNamespace(gru=gru), [leaves[0]], [leaves[1]], leaves[2], leaves[3],
                                          depths.view(V, 1, h, w), E[None], Kn[None], (h, w))
    gen = torch.Generator().manual_seed(78)
    ws = [torch.randn(o.shape, generator=gen) for o in out]
    sum((o * w_).sum() for o, w_ in zip(out, ws)).backward()
    save("ptf_small_grads.npz", w_latent=ws[0], w_xyz=ws[1], w_extrinsics=ws[2], w_depths=ws[3],
         d_latents=leaves[0].grad, d_coords=leaves[1].grad, d_densities=leaves[2].grad, d_weights=leaves[3].grad,
         **{"d_gru__" + k.replace(".", "__"): p_.grad for k, p_ in gru.named_parameters()})


if __name__ == "__main__":
    install_shim()
    if ONLY == {"backward"}:          # python make_golden.py backward: only the gradient fixtures
        gen_backward()
        sys.exit(0)
    if ONLY:
        gen_cost_volume()
        sys.exit(0)
    gen_ply()
    gen_depth_tail()
    gen_glue()
    gen_framing()
    gen_cost_volume()
    gen_ptf_and_adapter()
    gen_backward()
