"""Pins oracle/ptf_oracle.py to the reference's own fuse_gaussians outputs (golden vectors generated
by tests/golden/make_golden.py from /root/reference/src/model/encoder/encoder_freesplat.py:431-522),
including the output ORDER and the exact-tie case."""
import os

import numpy as np
import pytest
import torch

from oracle import ptf_oracle as po

HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    z = np.load(os.path.join(HERE, "golden", name))
    g = {k: torch.from_numpy(z[k]) if z[k].ndim else z[k].item() for k in z.files}
    gru = {k[len("gru__"):].replace("__", "."): v for k, v in g.items() if k.startswith("gru__")}
    return g, gru


@pytest.mark.parametrize("name", ["ptf_small.npz", "ptf_tie.npz"])
def test_fold_matches_reference(name):
    g, gru = load(name)
    h, w = int(g["h"]), int(g["w"])
    out = po.fuse_gaussians(gru, g["latents"], g["coords"], g["densities"], g["weights"], g["depths"],
                            g["extrinsics"][None], g["intrinsics"][None], (h, w))
    for got, key, tol in zip(out, ("out_latent", "out_xyz", "out_extrinsics", "out_depths"), (2e-5, 1e-5, 1e-5, 1e-5)):
        assert got.shape == g[key].shape, key          # same M  => same matches
        np.testing.assert_allclose(got.numpy(), g[key].numpy(), atol=tol, rtol=1e-5, err_msg=key)


def test_tie_fixture_really_has_ties():
    g, _ = load("ptf_tie.npz")
    # identical cameras + constant depth: every view-1 pixel fuses with exactly its view-0 twin
    assert g["out_latent"].shape[1] == int(g["h"]) * int(g["w"])


def test_match_step_edge_cases():
    h, w = 4, 6
    K = np.array([3.0, 3.0, 2.5, 1.5], np.float32)
    I4 = np.eye(4, dtype=np.float32)
    d = np.full(h * w, 2.0, np.float32)
    # empty state
    keep, fuse, fpix, app = po.match_step(np.zeros((0, 3), np.float32), I4, K, d, h, w)
    assert len(keep) == 0 and len(fuse) == 0 and len(app) == h * w
    # behind the camera / outside the frame / exact half-pixel (round half to even)
    xyz = np.array([[0, 0, -1.0], [100, 0, 2.0], [0.0, 0.0, 2.0], [(0.5 - 2.5) / 3 * 2, 0.0, 2.0]], np.float32)
    keep, fuse, fpix, app = po.match_step(xyz, I4, K, d, h, w)
    # point 2 -> px = 2.5 -> col 2 (half to even), py = 1.5 -> row 2 ; point 3 -> px 0.5 -> col 0
    assert list(fuse) == [2, 3] and list(fpix) == [2 * w + 2, 2 * w + 0]
    assert list(keep) == [0, 1] and len(app) == h * w - 2


def test_oracle_autograd_matches_reference_backward():
    """Gradients through the oracle's fold (torch autograd) against gradients through the REFERENCE'S OWN fuse_gaussians
    (tests/golden/ptf_small_grads.npz, make_golden.gen_backward: encoder_freesplat.py:431-522 with networks.py:188-214): latents,
    coordinates, densities, weights and the 12 GRU tensors, seeded weights on all four outputs."""
    g, gru = load("ptf_small.npz")
    z = np.load(os.path.join(HERE, "golden", "ptf_small_grads.npz"))
    gg = {k: torch.from_numpy(z[k]) for k in z.files}
    h, w = int(g["h"]), int(g["w"])
    params = {k: v.clone().requires_grad_(True) for k, v in gru.items()}
    leaves = [g[k].clone().requires_grad_(True) for k in ("latents", "coords", "densities", "weights")]
    out = po.fuse_gaussians(params, leaves[0], leaves[1], leaves[2], leaves[3], g["depths"], g["extrinsics"][None],
                            g["intrinsics"][None], (h, w))
    sum((o * gg[k]).sum() for o, k in zip(out, ("w_latent", "w_xyz", "w_extrinsics", "w_depths"))).backward()
    rel = lambda a, b: float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30)
    for t, key in zip(leaves, ("d_latents", "d_coords", "d_densities", "d_weights")):
        assert t.grad.shape == gg[key].shape and rel(t.grad, gg[key]) < 2e-5, (key, rel(t.grad, gg[key]))
    for k, p_ in params.items():
        want = gg["d_gru__" + k.replace(".", "__")]
        assert rel(p_.grad, want) < 5e-5, (k, rel(p_.grad, want))
