"""Pins oracle/adapter_oracle.py to the reference's GaussianAdapter outputs (golden vectors from
tests/golden/make_golden.py: /root/reference/src/model/encoder/common/gaussian_adapter.py)."""
import os

import numpy as np
import torch

from oracle import adapter_oracle as ao

HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    z = np.load(os.path.join(HERE, "golden", name))
    return {k: torch.from_numpy(z[k]) if z[k].ndim else z[k].item() for k in z.files}


def test_unproject_matches_reference():
    g = _load("ptf_small.npz")
    h, w = int(g["h"]), int(g["w"])
    V = g["depths"].shape[0]
    K0 = g["intrinsics"][0].clone()
    K0[0] *= w
    K0[1] *= h
    k0 = torch.stack([K0[0, 0], K0[1, 1], K0[0, 2], K0[1, 2]])
    xyz = ao.unproject(g["depths"].reshape(V, -1), g["extrinsics"], k0, h, w)
    np.testing.assert_allclose(xyz.numpy(), g["coords"][0, :, :, 0, 0].numpy(), atol=2e-6)


def test_gaussian_head_matches_reference():
    g = _load("adapter_small.npz")
    h, w = int(g["h"]), int(g["w"])
    M = g["extrinsics"].shape[0]
    mult = ao.scale_multiplier(g["intrinsics"], h, w)
    cov, sh, scales, rot = ao.gaussian_head(g["raw"].reshape(M, 34), g["depths"].reshape(M), g["extrinsics"], mult,
                                            g["sh_mask"])
    np.testing.assert_allclose(cov.numpy(), g["out_cov"].reshape(M, 3, 3).numpy(), rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(sh.numpy(), g["out_harmonics"].reshape(M, 3, 9).numpy(), rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(scales.numpy(), g["out_scales"].reshape(M, 3).numpy(), rtol=1e-6)
    np.testing.assert_allclose(rot.numpy(), g["out_rotations"].reshape(M, 4).numpy(), rtol=1e-6, atol=1e-8)
