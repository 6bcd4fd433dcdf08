"""Seeded synthetic "FreeSplat-like" inputs for the benchmark and the parity tests
(there are no datasets or checkpoints offline).

A scene is what FreeSplat's encoder would hand the decoder: one Gaussian per context pixel,
unprojected from V posed context cameras on a short arc in front of a noisy wall at 1-5 m, with
scales  U(0.5,15) * 0.1 * depth * (1/fx_px + 1/fy_px)  (gaussian_adapter.py:155-160,203-214 of
the reference), random rotations, opacity sigmoid(N(0,2)), SH degree 2 with the reference's
band mask (1, 0.025, 0.00625) (gaussian_adapter.py:127-133); then subsampled to exactly N.
seed 111123 = config/main.yaml:57 of the reference.
"""
from __future__ import annotations

import math

import numpy as np
import torch

SEED = 111123
# ScanNet-like colour camera (fx=fy=1170 px at 1296x968), normalised by (w, h)
FX_N, FY_N, CX_N, CY_N = 1170.0 / 1296.0, 1170.0 / 968.0, 0.5, 0.5


def intrinsics_normalized() -> np.ndarray:
    return np.array([[FX_N, 0, CX_N], [0, FY_N, CY_N], [0, 0, 1]], np.float32)


def _look_at_c2w(pos, target):
    """OpenCV camera (x right, y down, z forward) camera-to-world."""
    z = target - pos
    z = z / np.linalg.norm(z)
    up = np.array([0.0, -1.0, 0.0])
    x = np.cross(-up, z)  # y is down, so right = z x up... keep a right-handed frame
    x = x / np.linalg.norm(x)
    y = np.cross(z, x)
    c2w = np.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = x, y, z, pos
    return c2w


def arc_cameras(n: int, baseline: float = 0.25, jitter: float = 0.0, rng=None) -> np.ndarray:
    """n cameras spread over `baseline` metres along x, all looking at (0,0,3)."""
    out = []
    for i in range(n):
        t = 0.0 if n == 1 else i / (n - 1) - 0.5
        pos = np.array([t * baseline, 0.03 * math.sin(3.0 * t), 0.02 * t])
        if rng is not None and jitter > 0:
            pos = pos + rng.normal(0, jitter, 3)
        out.append(_look_at_c2w(pos, np.array([0.0, 0.0, 3.0])))
    return np.stack(out).astype(np.float32)


def wall_depth(u, v, rng, noise=0.02):
    """Smooth 1-5 m depth field over normalised image coords + N(0, noise)."""
    d = 2.6 + 0.9 * np.sin(2.1 * u + 0.3) * np.cos(1.7 * v - 0.2) + 0.5 * (u - 0.5) + 0.4 * np.sin(9.0 * v)
    d = np.clip(d, 1.0, 5.0)
    return d + rng.normal(0.0, noise, d.shape)


def make_scene(n_gaussians: int, n_context: int = 3, seed: int = SEED, sh_degree: int = 2,
               ctx_hw: tuple[int, int] = (384, 512), splat_scale: float = 1.0) -> dict:
    """Returns CPU float32 tensors: means [N,3], covariances [N,3,3], harmonics [N,3,d_sh],
    opacities [N], plus the context cameras.  `splat_scale`: every Gaussian's three axes multiplied by this factor (the same
    random draws: splat_scale = 1 is bit-identical to the scene every earlier number was measured on) -- see WORKLOADS."""
    rng = np.random.default_rng(seed)
    K = intrinsics_normalized()
    c2ws = arc_cameras(n_context)
    per_view = -(-n_gaussians // n_context)
    aspect = ctx_hw[1] / ctx_hw[0]
    gh = int(math.ceil(math.sqrt(per_view / aspect)))
    gw = int(math.ceil(per_view / gh))
    means, depths = [], []
    for c2w in c2ws:
        vs, us = np.meshgrid((np.arange(gh) + 0.5) / gh, (np.arange(gw) + 0.5) / gw, indexing="ij")
        d = wall_depth(us, vs, rng)
        x = (us - CX_N) / FX_N * d
        y = (vs - CY_N) / FY_N * d
        pc = np.stack([x, y, d, np.ones_like(d)], -1).reshape(-1, 4)
        means.append((pc @ c2w.T.astype(np.float64))[:, :3])
        depths.append(d.reshape(-1))
    means = np.concatenate(means)
    depths = np.concatenate(depths)
    keep = rng.permutation(means.shape[0])[:n_gaussians]
    keep.sort()  # keep the pixel-major order the encoder produces
    means, depths = means[keep], depths[keep]
    N = means.shape[0]
    fx_px, fy_px = FX_N * ctx_hw[1], FY_N * ctx_hw[0]
    mult = 0.1 * (1.0 / fx_px + 1.0 / fy_px)
    scales = rng.uniform(0.5, 15.0, (N, 3)) * depths[:, None] * mult * float(splat_scale)
    q = rng.normal(size=(N, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    x, y, z, w = q.T  # xyzw, as the reference's quaternion_to_matrix (gaussians.py:8-44)
    Rm = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                   2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                   2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(N, 3, 3)
    RS = Rm * scales[:, None, :]
    cov = RS @ RS.transpose(0, 2, 1)
    d_sh = (sh_degree + 1) ** 2
    mask = np.ones(d_sh)
    for deg in range(1, sh_degree + 1):
        mask[deg ** 2:(deg + 1) ** 2] = 0.1 * 0.25 ** deg
    sh = rng.normal(size=(N, 3, d_sh))
    sh[:, :, 0] = rng.uniform(-1.0, 1.0, (N, 3))
    sh = sh * mask
    opac = 1.0 / (1.0 + np.exp(-rng.normal(0.0, 2.0, N)))
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return dict(means=f(means), covariances=f(cov), harmonics=f(sh), opacities=f(opac),
                context_extrinsics=f(c2ws), intrinsics=f(K))


def target_cameras(n: int, seed: int = SEED) -> dict:
    """n target views interpolated along (and slightly off) the context arc; near 0.5 / far 15."""
    rng = np.random.default_rng(seed + 1)
    c2w = arc_cameras(n, baseline=0.2, jitter=0.01, rng=rng)
    K = np.broadcast_to(intrinsics_normalized(), (n, 3, 3)).copy()
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return dict(extrinsics=f(c2w), intrinsics=f(K), near=torch.full((n,), 0.5), far=torch.full((n,), 15.0))


WORKLOADS = {
    # name: (H, W, N gaussians)  -- BASELINE.json configs
    "c1_256x256_plumbing": (256, 256, 20_000),
    "c2_640x480_300k": (480, 640, 300_000),
    "c3_968x1296_1M": (968, 1296, 1_000_000),
    # the UNFRIENDLY config-3 workload: the same 1.0 M Gaussians and cameras with every splat at 2.5x its
    # size on screen -- what a camera at 0.4x the distance sees of surfaces whose Gaussians all stay in frame.  (Moving the
    # target camera alone does not do it: a FreeSplat scene holds one Gaussian per context pixel, so at 1/3 of the distance
    # each splat is 3x larger but only 1/4 of them stay in view -- the oracle counts 9.2 -> 7.1 list entries per Gaussian.)
    # Oracle (3-sigma rectangles, no tile culling) at 2.0x: 25.5 entries per Gaussian, median tile list 5 316, EVERY tile list
    # > 2 048 entries, i.e. the global-memory sort path of sort_blend_kernel (include/freesplat_amd.h) on every tile.
    "c3_closeup_968x1296_1M": (968, 1296, 1_000_000),
}
SCENE_OPTIONS = {"c3_closeup_968x1296_1M": {"splat_scale": 2.5}}


def workload_scene(name: str) -> dict:
    """The scene of a named workload (make_scene with its options)."""
    return make_scene(WORKLOADS[name][2], **SCENE_OPTIONS.get(name, {}))
