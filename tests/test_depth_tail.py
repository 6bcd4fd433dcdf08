"""N3: depth-regression tail.  CPU: the oracle against the reference's golden vectors (real DepthDecoder
logits).  GPU: the fused HIP op against the golden vectors and the oracle's autograd."""
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    z = np.load(os.path.join(HERE, "golden", name))
    return {k: torch.from_numpy(z[k]) if z[k].ndim else z[k].item() for k in z.files}


@pytest.mark.parametrize("name", ["depth_tail_log.npz", "depth_tail_inv.npz"])
def test_oracle_matches_reference(name):
    from oracle.depth_tail_oracle import depth_tail
    g = _load(name)
    lp = bool(g["log_planes"])
    o = depth_tail(g["logits"], g["candidates"], lp)
    for k in ("coarse", "depth", "depth_map", "depth_weights"):
        assert torch.equal(o[k], g[k]), k
    m = depth_tail(g["module_logits"], g["candidates"], lp)            # the module's own forward outputs
    assert torch.equal(m["coarse"], g["module_log_depth"]) and torch.equal(m["depth_map"], g["module_depth_map"])
    assert torch.equal(m["depth_weights"], g["module_depth_weights"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["depth_tail_log.npz", "depth_tail_inv.npz"])
def test_hip_matches_reference_and_autograd(hip_device, name):
    from oracle.depth_tail_oracle import depth_tail
    from freesplat_amd.depth_tail import depth_regression_tail
    g = _load(name)
    lp = bool(g["log_planes"])
    lg = g["logits"].to(hip_device).requires_grad_(True)
    o = depth_regression_tail(lg, g["candidates"].to(hip_device), lp)
    for k, tol in (("coarse", 2e-6), ("depth", 2e-5), ("depth_map", 2e-5), ("depth_weights", 2e-6)):
        scale = g[k].abs().max().item()
        assert (o[k].detach().cpu() - g[k]).abs().max().item() <= tol * max(scale, 1.0), k
    gen = torch.Generator().manual_seed(1)
    gs = {k: torch.randn(g[k].shape, generator=gen) for k in ("coarse", "depth", "depth_map", "depth_weights")}
    sum((o[k] * gs[k].to(hip_device)).sum() for k in gs).backward()
    lc = g["logits"].double().clone().requires_grad_(True)
    r = depth_tail(lc, g["candidates"].double(), lp)
    sum((r[k] * gs[k].double()).sum() for k in gs).backward()
    s = lc.grad.abs().max().item()
    e = (lg.grad.cpu().double() - lc.grad).abs()
    # arg-max ties / near-ties of the upsampled probabilities may pick a different plane at isolated pixels
    assert e.flatten().kthvalue(int(e.numel() * 0.999)).values.item() <= 2e-4 * s and e.mean().item() <= 1e-5 * s


def test_backward_gather_window_covers_every_tap():
    """CPU check of the assumption behind fs_depth_tail_backward's gather (csrc/depth_tail.hip:depth_tail_bwd_kernel): a fine
    pixel f of the x2 align_corners upsampling takes its taps from coarse rows floor(f (n - 1) / (2 n - 1)) and + 1, computed
    in float32 exactly as bilin_x2 does; coarse pixel Y looks for its contributors among the fine rows 2 Y - 2 .. 2 Y + 3.
    Every tap with a non-zero weight of every fine index must fall inside that window, for every size up to 1500 (968 x 1296
    images have 484 x 648 logits) -- per axis, which is all the separable kernel needs."""
    for n in list(range(1, 130)) + [192, 256, 324, 484, 648, 1023, 1500]:
        f = np.arange(2 * n, dtype=np.float32)
        r = np.float32(n - 1) / np.float32(2 * n - 1) if n > 1 else np.float32(0)
        src = f * r
        y0 = np.minimum(src.astype(np.int32), n - 1)
        y1 = np.minimum(y0 + 1, n - 1)
        fy = src - y0.astype(np.float32)
        for taps, wts in ((y0, 1.0 - fy), (y1, fy)):
            live = wts != 0
            fi = np.arange(2 * n)[live]
            Y = taps[live]
            assert np.all(fi >= 2 * Y - 2) and np.all(fi <= 2 * Y + 3), n


@pytest.mark.gpu
@pytest.mark.parametrize("B,D,h2,w2,log_planes,which", [
    (1, 7, 5, 9, True, "all"),            # D not a multiple of the four plane classes, a ragged 64-pixel workgroup
    (2, 130, 13, 21, True, "all"),        # D > 128: the backward walks two plane chunks
    (1, 300, 6, 11, False, "all"),        # three chunks, inverse-depth planes
    (3, 32, 1, 1, True, "all"),           # one coarse pixel per image: every fine pixel taps it
    (2, 64, 17, 40, True, "weights"),     # only depth_weights has a gradient (no bilinear-map term)
    (2, 64, 17, 40, False, "map"),        # only the x2 map
    (2, 64, 17, 40, True, "coarse"),      # no gradient through the upsampled outputs at all
])
def test_hip_forward_and_backward_on_seeded_shapes(hip_device, B, D, h2, w2, log_planes, which):
    """Forward outputs and the gradient of the logits against the float64 oracle's autograd on shapes the goldens do not
    cover (fs_depth_tail_forward: 4 plane classes x batches of 8; fs_depth_tail_backward: the 6 x 6 gather of fine pixels
    per coarse pixel, plane chunks of 128, every subset of output gradients the C ABI allows).  Fine pixels whose arg max
    over the planes is a near-tie may pick another plane: such pixels are masked out of the depth_weights gradient."""
    from oracle.depth_tail_oracle import depth_tail
    from freesplat_amd.depth_tail import depth_regression_tail
    gen = torch.Generator().manual_seed(B * 1000 + D)
    logits = 2.0 * torch.randn(B, D, h2, w2, generator=gen)
    lo, hi = (0.5, 15.0)
    cand = (torch.log(torch.tensor(lo)) + torch.linspace(0, 1, D) * torch.log(torch.tensor(hi / lo))) if log_planes \
        else (1.0 / hi + torch.linspace(0, 1, D) * (1.0 / lo - 1.0 / hi))
    lg = logits.to(hip_device).requires_grad_(True)
    o = depth_regression_tail(lg, cand.to(hip_device), log_planes)
    lc = logits.double().clone().requires_grad_(True)
    r = depth_tail(lc, cand.double(), log_planes)
    for k in ("coarse", "depth", "depth_map", "depth_weights"):
        assert (o[k].detach().cpu().double() - r[k].detach()).abs().max().item() <= 2e-5 * max(1.0, r[k].abs().max().item()), k
    # near-ties of the upsampled probabilities: the two largest planes within 1e-5 of each other
    up = torch.nn.functional.interpolate(torch.softmax(logits.double(), 1), scale_factor=2, mode="bilinear", align_corners=True)
    top2 = up.topk(2, dim=1).values if D > 1 else None
    clear = ((top2[:, 0] - top2[:, 1]) > 1e-5)[:, None] if D > 1 else torch.ones_like(r["depth_weights"], dtype=torch.bool)
    gs = {k: torch.randn(r[k].shape, generator=gen) for k in ("coarse", "depth", "depth_map", "depth_weights")}
    gs["depth_weights"] = gs["depth_weights"] * clear
    keys = {"all": ("coarse", "depth", "depth_map", "depth_weights"), "weights": ("depth_weights",), "map": ("depth_map",),
            "coarse": ("coarse", "depth")}[which]
    sum((o[k] * gs[k].to(hip_device)).sum() for k in keys).backward()
    sum((r[k] * gs[k].double()).sum() for k in keys).backward()
    e = (lg.grad.cpu().double() - lc.grad).abs()
    assert e.max().item() <= 2e-5 * lc.grad.abs().max().item(), (e.max().item(), lc.grad.abs().max().item())


@pytest.mark.gpu
def test_hip_native_size_and_coarse_only(hip_device):
    from oracle.depth_tail_oracle import depth_tail
    from freesplat_amd.depth_tail import apply_to_depth_outputs, depth_regression_tail
    gen = torch.Generator().manual_seed(5)
    B, D, h2, w2 = 2, 128, 192, 256
    logits = 3.0 * torch.randn(B, D, h2, w2, generator=gen)
    cand = torch.log(torch.tensor(0.5)) + torch.linspace(0, 1, D) * torch.log(torch.tensor(30.0))
    ref = depth_tail(logits, cand, True)
    with torch.no_grad():
        o = depth_regression_tail(logits.to(hip_device), cand.to(hip_device), True)
        c = depth_regression_tail(logits.to(hip_device), cand.to(hip_device), True, upsample=False)
    for k in ("coarse", "depth", "depth_map", "depth_weights"):
        assert (o[k].cpu() - ref[k]).abs().max().item() <= 5e-5 * max(1.0, ref[k].abs().max().item()), k
    assert torch.equal(c["coarse"], o["coarse"]) and "depth_map" not in c
    outs = apply_to_depth_outputs({}, {0: logits.to(hip_device), 1: logits[:, :, ::2, ::2].contiguous().to(hip_device)},
                                  cand.to(hip_device))
    assert set(outs) == {"depth_pred_s0_b1hw", "log_depth_pred_s0_b1hw", "depth_pred_s1_b1hw", "log_depth_pred_s1_b1hw",
                         "depth_pred_s-1_b1hw", "depth_weights"}


@pytest.mark.gpu
def test_depth_decoder_forward_dropin(hip_device):
    """depth_decoder_forward on a stand-in with the attributes of the reference DepthDecoder (max_depth 2 here)
    against the same pyramid evaluated op by op with torch and the oracle tail."""
    import torch.nn as nn
    import torch.nn.functional as F
    from oracle.depth_tail_oracle import depth_tail
    from freesplat_amd.depth_tail import depth_decoder_forward
    torch.manual_seed(3)
    D, ch = 16, [8, 12, 16]

    class Stub(nn.Module):
        def __init__(s):
            super().__init__()
            s.max_depth, s.log_planes = 2, True
            s.convs = nn.ModuleDict()
            for j in range(1, 3):
                for i in range(2 - j, -1, -1):
                    s.convs[f"right_conv_{i}{j - 1}"] = nn.Conv2d(ch[i], 8, 3, padding=1)
                    s.convs[f"diag_conv_{i + 1}{j - 1}"] = nn.Conv2d(ch[i + 1] if j == 1 else 8, 8, 3, padding=1)
                    n = 16
                    if i + j != 2:
                        s.convs[f"up_conv_{i + 1}{j}"] = nn.Conv2d(8, 8, 3, padding=1)
                        n = 24
                    s.convs[f"in_conv_{i}{j}"] = nn.Conv2d(n, 8, 3, padding=1)
                    s.convs[f"output_{i}"] = nn.Conv2d(8, 8, 1)
            s.conv_depth = nn.ModuleDict({f"{i}": nn.Conv2d(8, D, 1) for i in range(2)})
            s.conv_last = nn.Conv2d(8, 8, 1)
            s.depth_candi_curr = (torch.log(torch.tensor(0.5)) + torch.linspace(0, 1, D) * torch.log(torch.tensor(20.0))
                                  ).view(1, D, 1, 1)

    m = Stub().to(hip_device)
    # level-0 inputs of step j=2 come from the j=1 outputs (8 channels)
    m.convs["right_conv_01"] = nn.Conv2d(8, 8, 3, padding=1).to(hip_device)
    feats = [torch.randn(2, ch[i], 32 >> i, 48 >> i, device=hip_device) for i in range(3)]
    with torch.no_grad():
        got = depth_decoder_forward(m, feats)
        up = lambda x: F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
        o11 = m.convs["in_conv_11"](torch.cat([m.convs["right_conv_10"](feats[1]), up(m.convs["diag_conv_20"](feats[2]))], 1))
        o01 = m.convs["in_conv_01"](torch.cat([m.convs["right_conv_00"](feats[0]), up(m.convs["diag_conv_10"](feats[1])),
                                               up(m.convs["up_conv_11"](o11))], 1))
        o02 = m.convs["in_conv_02"](torch.cat([m.convs["right_conv_01"](o01), up(m.convs["diag_conv_11"](o11))], 1))
        p1, p0 = m.convs["output_1"](o11), m.convs["output_0"](o02)
        t1 = depth_tail(m.conv_depth["1"](p1).cpu(), m.depth_candi_curr.view(-1).cpu(), True, upsample=False)
        t0 = depth_tail(m.conv_depth["0"](p0).cpu(), m.depth_candi_curr.view(-1).cpu(), True)
    close = lambda a, b: (a.cpu() - b.cpu()).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item())
    assert close(got["output_pred_s0_b1hw"], p0) and close(got["output_pred_s1_b1hw"], p1)
    assert close(got["depth_pred_s1_b1hw"], t1["depth"]) and close(got["log_depth_pred_s1_b1hw"], t1["coarse"])
    assert close(got["depth_pred_s0_b1hw"], t0["depth"]) and close(got["depth_pred_s-1_b1hw"], t0["depth_map"])
    assert close(got["depth_weights"], t0["depth_weights"])
    assert close(got["output_pred_s-1_b1hw"], m.conv_last(up(p0)))
