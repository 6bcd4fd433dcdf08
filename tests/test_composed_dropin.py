"""The composed drop-in on the GPU: `encoder_forward` (what compat.patch_reference() binds as
`EncoderFreeSplat.forward`, encoder_freesplat.py:196-429) driving the HIP cost volume, HIP depth-regression tail, HIP
unprojection, HIP PTF fold and HIP Gaussian head, then `DecoderSplattingCUDA` (decoder_splatting_cuda.py:35-75) and an MSE
loss -- FORWARD AND BACKWARD, b = 1 and b = 2, at BASELINE config 1's size (2 views, 256 x 256).

The out-of-scope modules of the encoder (backbone, cv_encoder, the depth decoder's convolution trunk,
high_resolution_skip, to_gaussians) are small deterministic torch stand-ins, identical on both sides.  The reference
side is the SAME `encoder_forward` (pinned entry by entry to the reference's own forward on the reference's modules by
tests/test_compat_reference.py) over an encoder whose hot-path modules are the reference-pinned ORACLES, on the CPU, and
the oracle rasterizer behind a torch autograd wrapper.  Compared: every entry of the result dictionary, the Gaussians,
the rendered images, the loss, and the gradient of the loss w.r.t. EVERY parameter of the stand-ins and of the hot-path
modules (cost-volume MLP, GRU) -- i.e. the whole chain of backward kernels composed.
"""
import os
import sys
import types

import numpy as np
import pytest
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

H = W = 256
V, D, C = 2, 16, 48
NEAR, FAR = 0.5, 15.0


def _up2(x):
    return torch.nn.functional.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)


class _Backbone(nn.Module):
    def __init__(self):
        super().__init__()
        self.c0 = nn.Conv2d(3, 8, 3, stride=2, padding=1)
        self.c1 = nn.Conv2d(8, C, 3, stride=2, padding=1)

    def forward(self, x):
        f0 = torch.tanh(self.c0(x))
        return [f0, self.c1(f0)]


class _CVEncoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.c = nn.Conv2d(D + C, 24, 3, padding=1)

    def forward(self, volume, feats):
        return [torch.tanh(self.c(torch.cat([volume, feats[0]], 1)))]


class _DepthDecoder(nn.Module):
    """Convolution trunk stand-in + the real regression tail (`tail`: the HIP op or its oracle)."""

    def __init__(self, tail):
        super().__init__()
        self.max_depth = 1
        self.tail = tail
        self.conv_depth = nn.Conv2d(8 + 24, D, 3, padding=1)
        self.conv_last = nn.Conv2d(8 + 24, 1 + 64, 3, padding=1)
        self.register_buffer("cand", torch.linspace(np.log(1.2), np.log(2.6), D))

    def forward(self, feats):
        x = torch.cat([feats[0], _up2(feats[1])], 1)
        r = self.tail(self.conv_depth(x), self.cand)
        return {"depth_pred_s0_b1hw": r["depth"], "log_depth_pred_s0_b1hw": r["coarse"], "depth_pred_s-1_b1hw": r["depth_map"],
                "depth_weights": r["depth_weights"], "output_pred_s-1_b1hw": self.conv_last(_up2(x))}


class _OracleAdapter(nn.Module):
    """GaussianAdapter's interface (gaussian_adapter.py:135-201) over oracle/adapter_oracle.py."""

    def forward(self, extrinsics, intrinsics, coordinates, depths, opacities, raw_gaussians, image_shape, eps=1e-8,
                fusion=False, coords=None):
        from freesplat_amd.gaussian_adapter import Gaussians
        from oracle import adapter_oracle as ao
        h, w = image_shape
        if fusion:
            out = []
            for i in range(intrinsics.shape[0]):
                K = intrinsics[i, 0].reshape(3, 3)
                k0 = torch.stack([K[0, 0] * w, K[1, 1] * h, K[0, 2] * w, K[1, 2] * h])
                v = intrinsics.shape[1]
                out.append(ao.unproject(depths[i].reshape(v, h * w), extrinsics[i].reshape(v, 4, 4), k0, h, w))
            return torch.stack(out)[:, :, :, None, None, :]
        lead = opacities.shape
        M = opacities.numel()
        mult = ao.scale_multiplier(intrinsics, h, w)
        mult = mult.expand(lead).reshape(M) if mult.numel() > 1 else mult.reshape(())
        mask = torch.tensor([1.0] + [0.025] * 3 + [0.00625] * 5)
        cov, sh, scales, rot = ao.gaussian_head(raw_gaussians.expand(*lead, raw_gaussians.shape[-1]).reshape(M, -1),
                                                depths.expand(lead).reshape(M), extrinsics.expand(*lead, 4, 4).reshape(M, 4, 4),
                                                mult, mask)
        return Gaussians(means=coords, covariances=cov.reshape(*lead, 3, 3), harmonics=sh.reshape(*lead, 3, 9),
                         opacities=opacities, scales=scales.reshape(*lead, 3), rotations=rot.reshape(*lead, 4))


class _Encoder(nn.Module):
    """The attributes `encoder_forward` reads from an EncoderFreeSplat (encoder_freesplat.py:100-188)."""

    def __init__(self, oracle: bool):
        super().__init__()
        from freesplat_amd.cost_volume import AVGFeatureVolumeManager
        from freesplat_amd.ptf import GRU
        self.cfg = types.SimpleNamespace(num_views=V, num_surfaces=1)
        self.max_depth = 1
        torch.manual_seed(11)
        self.backbone = _Backbone()
        self.cv_encoder = _CVEncoder()
        self.high_resolution_skip = nn.ModuleList([nn.Conv2d(3, 64, 3, padding=1)])
        self.to_gaussians = nn.Sequential(nn.ReLU(), nn.Linear(64, 36))
        self.cost_volume = AVGFeatureVolumeManager(H // 4, W // 4, num_depth_bins=D, mlp_channels=[202, 32, 32, 1],
                                                   matching_dim_size=C)
        self.gru = GRU()
        if oracle:
            from oracle import depth_tail_oracle as dto
            self.depth_decoder = _DepthDecoder(lambda lg, cd: dto.depth_tail(lg, cd, True, True))
            self.gaussian_adapter = _OracleAdapter()
        else:
            from freesplat_amd.depth_tail import depth_regression_tail
            from freesplat_amd.gaussian_adapter import GaussianAdapter, GaussianAdapterCfg
            self.depth_decoder = _DepthDecoder(lambda lg, cd: depth_regression_tail(lg, cd, True, True))
            self.gaussian_adapter = GaussianAdapter(GaussianAdapterCfg(0.5, 15.0, 2))
        self.oracle = oracle

    def fuse_gaussians(self, *a, **k):
        if not self.oracle:
            from freesplat_amd.ptf import fuse_gaussians
            return fuse_gaussians(self, *a, **k)
        from oracle import ptf_oracle as po
        lat, coords = a[0][0], a[1][0]          # (the reference passes one-element lists, encoder_freesplat.py:364-368)
        return po.fuse_gaussians({k_: v for k_, v in self.gru.state_dict(keep_vars=True).items()}, lat, coords, *a[2:], **k)


def _oracle_cost_volume(m):
    """The module's forward over oracle/cost_volume_oracle.py (same weights: autograd reaches m.mlp's parameters)."""
    from oracle import cost_volume_oracle as cvo

    def fwd(cur_feats, src_feats, src_extrinsics, src_poses, src_Ks, cur_invK, min_depth, max_depth, **kw):
        net = m.mlp.net
        mlp = dict(w1=net[0].weight, b1=net[0].bias, w2=net[2].weight, b2=net[2].bias, w3=net[4].weight, b3=net[4].bias)
        return cvo.cost_volume(cur_feats, src_feats, src_extrinsics, src_Ks, cur_invK, min_depth, max_depth, D, mlp)
    return fwd


class _OracleRaster(torch.autograd.Function):
    """oracle/raster_oracle.c forward + backward of ONE view as a torch op (colour only)."""

    @staticmethod
    def forward(ctx, means, cov6, shs, opac, frame):
        from oracle import raster_oracle as ro
        tx, ty, view, full, campos = frame
        st = ro.forward(H, W, tx, ty, np.zeros(3, np.float32), view, full, 2, campos, means.detach().numpy(),
                        cov6.detach().numpy(), opac.detach().numpy(), shs=shs.detach().numpy())
        ctx.st = st
        return torch.from_numpy(st["color"].copy())

    @staticmethod
    def backward(ctx, g):
        from oracle import raster_oracle as ro
        r = ro.backward(ctx.st, g.contiguous().numpy())
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        return t(r["means3D"]), t(r["cov3D"]), t(r["shs"]), t(r["opacities"]).reshape(-1), None


def _render_oracle(gs, tgt_E, tgt_K, frames):
    """DecoderSplattingCUDA.forward's arithmetic (decoder_splatting_cuda.py:48-62, cuda_splatting.py:64-78) over the oracle
    rasterizer: [b, v, 3, H, W].  `frames` = fs_frame_views' matrices (the product's own framing: tests/test_pipeline_c1.py
    explains why image comparisons must share it; the framing itself is pinned to the reference elsewhere)."""
    r, c = torch.triu_indices(3, 3)
    out = []
    for i in range(gs.means.shape[0]):
        views = []
        for j in range(tgt_E.shape[1]):
            campos, scale, tanfov, view, full = (t[i * tgt_E.shape[1] + j] for t in frames)
            s = scale
            fr = (float(tanfov[0]), float(tanfov[1]), view.numpy(), full.numpy(), campos.numpy())
            views.append(_OracleRaster.apply(gs.means[i] * s, (gs.covariances[i] * s * s)[:, r, c].contiguous(),
                                             gs.harmonics[i].transpose(-1, -2).contiguous(), gs.opacities[i], fr))
        out.append(torch.stack(views))
    return torch.stack(out)


def _context(b, dev=None):
    import inputs
    g = torch.Generator().manual_seed(99)
    cams = [inputs.cameras(V, H, W, baseline=0.3, seed=5 + i) for i in range(b)]
    ctx = {"image": torch.rand(b, V, 3, H, W, generator=g), "extrinsics": torch.stack([c_[0] for c_ in cams]),
           "intrinsics": torch.stack([c_[1] for c_ in cams]), "near": torch.full((b, V), NEAR), "far": torch.full((b, V), FAR)}
    tgt = torch.stack([inputs.cameras(2, H, W, baseline=0.2, seed=9 + i)[0] for i in range(b)])
    target = torch.rand(b, 2, 3, H, W, generator=g)
    mv = (lambda t: t.to(dev)) if dev is not None else (lambda t: t)
    return {k: mv(v) for k, v in ctx.items()}, mv(tgt), mv(target)


def _grads(enc):
    return {n: p.grad.detach().cpu().clone() for n, p in enc.named_parameters() if p.grad is not None}


@pytest.mark.gpu
@pytest.mark.parametrize("b", [1, 2])
def test_composed_dropin_forward_backward_vs_oracle_chain(hip_device, b):
    from freesplat_amd.decoder import DecoderSplattingCUDA, frame_views
    from freesplat_amd.encoder_forward import encoder_forward
    dev = hip_device
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    # ---- product: HIP modules behind encoder_forward + DecoderSplattingCUDA ----
    enc = _Encoder(oracle=False).to(dev)
    ctx, tgt_E, target = _context(b, dev)
    res = encoder_forward(enc, dict(ctx), 0)
    dec = DecoderSplattingCUDA((0.0, 0.0, 0.0)).to(dev)
    n_t = tgt_E.shape[1]
    tgt_K = ctx["intrinsics"][:, :1].expand(b, n_t, 3, 3).contiguous()
    imgs = []
    for i in range(b):     # (the reference renders scene by scene when the scenes' Gaussian counts differ: model_wrapper.py:236-250)
        o = dec(res["gaussians"][i], tgt_E[i:i + 1], tgt_K[i:i + 1], torch.full((1, n_t), NEAR, device=dev),
                torch.full((1, n_t), FAR, device=dev), (H, W), depth_mode=None)
        imgs.append(o.color)
    img = torch.cat(imgs)
    loss = ((img - target) ** 2).mean()
    loss.backward()
    g_hip = _grads(enc)
    # ---- oracle chain: the same encoder_forward over the oracles, CPU ----
    ref_enc = _Encoder(oracle=True)
    ref_enc.load_state_dict(enc.state_dict())
    ref_enc.cost_volume.forward = _oracle_cost_volume(ref_enc.cost_volume)
    ctx_c, tgt_c, target_c = _context(b)
    ref = encoder_forward(ref_enc, dict(ctx_c), 0)
    frames = [t.cpu() for t in frame_views(tgt_E.reshape(b * n_t, 4, 4), tgt_K.reshape(b * n_t, 3, 3),
                                           torch.full((b * n_t,), NEAR, device=dev), torch.full((b * n_t,), FAR, device=dev), True)]
    ref_imgs = []
    for i in range(b):
        g = ref["gaussians"][i]
        one = types.SimpleNamespace(means=g.means, covariances=g.covariances, harmonics=g.harmonics, opacities=g.opacities)
        ref_imgs.append(_render_oracle(one, tgt_c[i:i + 1], None, [t[i * n_t:(i + 1) * n_t] for t in frames]))
    ref_img = torch.cat(ref_imgs)
    ref_loss = ((ref_img - target_c) ** 2).mean()
    ref_loss.backward()
    g_ref = _grads(ref_enc)
    # ---- forward: dictionary entries, Gaussians, images ----
    assert res["num_gaussians"] < V * H * W                                                        # something fused
    for k in ("depth_num0_s-1", "depth_num0_s0", "depth_num0_s-1_b1hw", "depth_num0_s0_b1hw"):
        assert (res[k].detach().cpu() - ref[k].detach()).abs().max().item() <= 2e-5, k
    # The fold's decisions are discrete (round-half-even pixel, depth test): fed with depths that agree to ~1e-6 the two
    # chains may decide one borderline pixel in ~10^5 differently, which shifts every later row.  Same count -> compare
    # the Gaussians row by row; otherwise at most two borderline pixels apart and only images / loss / gradients compared.
    same_fold = all(res["gaussians"][i].means.shape == ref["gaussians"][i].means.shape for i in range(b))
    for i in range(b):
        assert abs(res["gaussians"][i].means.shape[1] - ref["gaussians"][i].means.shape[1]) <= 2
        for f, tol in (("means", 2e-5), ("covariances", 1e-6), ("harmonics", 1e-5), ("opacities", 1e-5)):
            a, r_ = getattr(res["gaussians"][i], f).detach().cpu(), getattr(ref["gaussians"][i], f).detach()
            if same_fold:
                assert (a - r_).abs().max().item() <= tol, (i, f, (a - r_).abs().max().item())
    # images: ~10^5 near-coplanar Gaussians (one per pixel of two views of the same surface) -- a depth-order flip between
    # two of them moves a pixel by up to a few percent (tests/test_pipeline_c1.py); everything else agrees to ~1e-6
    err = (img.detach().cpu() - ref_img.detach()).abs().flatten()
    p999 = float(err.kthvalue(int(0.999 * err.numel())).values)
    assert float(err.mean()) <= 2e-5 and p999 <= 2e-3 and float(err.max()) <= 0.1, (float(err.max()), p999, float(err.mean()))
    assert abs(float(loss.detach()) - float(ref_loss.detach())) <= 1e-5 * max(1.0, abs(float(ref_loss.detach()))), (float(loss.detach()), float(ref_loss.detach()))
    # ---- backward: every parameter of the chain ----
    assert set(g_hip) == set(g_ref) and len(g_ref) >= 30
    bad = []
    for n in sorted(g_ref):
        a, r_ = g_hip[n].double().flatten(), g_ref[n].double().flatten()
        cos = float((a @ r_) / (a.norm() * r_.norm() + 1e-300))
        rel = float((a - r_).norm() / (r_.norm() + 1e-300))
        if not (cos >= 0.9995 and rel <= 3e-2):
            bad.append((n, round(cos, 6), round(rel, 5)))
    assert not bad, bad
