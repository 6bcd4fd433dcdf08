"""On-disk Gaussian format: drop-in for the reference's
/root/reference/src/model/ply_export.py:26-92 `export_ply(extrinsics, means, scales, rotations, harmonics,
opacities, path)` -- same scene normalisation (median shift, 95 % quantile rescale), same viewer rotation
(+Z up, -45 deg about Z, camera-space default view), rotations re-expressed as wxyz quaternions, DC band
only, log-scales -- written as a binary little-endian PLY with the 3DGS attribute names
(x y z nx ny nz f_dc_0..2 opacity scale_0..2 rot_0..3).  Pure torch/numpy (no plyfile / scipy needed).
Host-side I/O, not a kernel.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch
from torch import Tensor


def construct_list_of_attributes(num_rest: int) -> list[str]:
    attributes = ["x", "y", "z", "nx", "ny", "nz"]
    attributes += [f"f_dc_{i}" for i in range(3)]
    attributes += [f"f_rest_{i}" for i in range(num_rest)]
    attributes.append("opacity")
    attributes += [f"scale_{i}" for i in range(3)]
    attributes += [f"rot_{i}" for i in range(4)]
    return attributes


def _quat_xyzw_to_matrix(q: np.ndarray) -> np.ndarray:
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                     2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                     2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)


def _matrix_to_quat_xyzw(m: np.ndarray) -> np.ndarray:
    """Rotation matrices [G,3,3] -> unit quaternions xyzw (largest-component branch, numerically stable)."""
    t = np.stack([1 + m[:, 0, 0] - m[:, 1, 1] - m[:, 2, 2], 1 - m[:, 0, 0] + m[:, 1, 1] - m[:, 2, 2],
                  1 - m[:, 0, 0] - m[:, 1, 1] + m[:, 2, 2], 1 + m[:, 0, 0] + m[:, 1, 1] + m[:, 2, 2]], -1)
    k = t.argmax(-1)
    q = np.empty((m.shape[0], 4))
    for c in range(4):
        s = k == c
        if not s.any():
            continue
        mm, tt = m[s], np.sqrt(np.maximum(t[s, c], 1e-30)) * 2
        if c == 0:
            q[s] = np.stack([tt / 4, (mm[:, 0, 1] + mm[:, 1, 0]) / tt, (mm[:, 0, 2] + mm[:, 2, 0]) / tt, (mm[:, 2, 1] - mm[:, 1, 2]) / tt], -1)
        elif c == 1:
            q[s] = np.stack([(mm[:, 0, 1] + mm[:, 1, 0]) / tt, tt / 4, (mm[:, 1, 2] + mm[:, 2, 1]) / tt, (mm[:, 0, 2] - mm[:, 2, 0]) / tt], -1)
        elif c == 2:
            q[s] = np.stack([(mm[:, 0, 2] + mm[:, 2, 0]) / tt, (mm[:, 1, 2] + mm[:, 2, 1]) / tt, tt / 4, (mm[:, 1, 0] - mm[:, 0, 1]) / tt], -1)
        else:
            q[s] = np.stack([(mm[:, 2, 1] - mm[:, 1, 2]) / tt, (mm[:, 0, 2] - mm[:, 2, 0]) / tt, (mm[:, 1, 0] - mm[:, 0, 1]) / tt, tt / 4], -1)
    return q / np.linalg.norm(q, axis=-1, keepdims=True)


def ply_attributes(extrinsics: Tensor, means: Tensor, scales: Tensor, rotations: Tensor, harmonics: Tensor,
                   opacities: Tensor) -> np.ndarray:
    """The [G,17] float32 attribute table export_ply writes (ply_export.py:35-88)."""
    means = means - means.median(dim=0).values                                   # :36
    scale_factor = means.abs().quantile(0.95, dim=0).max()                       # :39
    means = means / scale_factor
    scales = scales / scale_factor
    rotation = torch.tensor([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], dtype=torch.float32, device=means.device)   # :44-49
    a = np.deg2rad(-45.0)                                                        # :54-58 rotvec (0,0,-45 deg)
    adjustment = torch.tensor([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], dtype=torch.float32,
                              device=means.device)
    rotation = adjustment @ rotation
    rotation = rotation @ extrinsics[:3, :3].inverse()                           # :63
    means = torch.einsum("ij,gj->gi", rotation, means)                          # :66
    rm = rotation.detach().cpu().double().numpy() @ _quat_xyzw_to_matrix(rotations.detach().cpu().double().numpy())
    q = _matrix_to_quat_xyzw(rm)                                                 # :69-71
    wxyz = np.stack((q[:, 3], q[:, 0], q[:, 1], q[:, 2]), -1)                    # :72-73
    n = lambda t: t.detach().cpu().float().numpy()
    return np.concatenate((n(means), np.zeros((means.shape[0], 3), np.float32), n(harmonics[..., 0]),
                           n(opacities[..., None]), n(scales.log()), wxyz.astype(np.float32)), axis=1).astype(np.float32)


def export_ply(extrinsics: Tensor, means: Tensor, scales: Tensor, rotations: Tensor, harmonics: Tensor,
               opacities: Tensor, path: Path) -> None:
    table = ply_attributes(extrinsics, means, scales, rotations, harmonics, opacities)
    names = construct_list_of_attributes(0)
    assert table.shape[1] == len(names)
    path = Path(path)
    path.parent.mkdir(exist_ok=True, parents=True)
    header = "ply\nformat binary_little_endian 1.0\n" + f"element vertex {table.shape[0]}\n" + \
             "".join(f"property float {a}\n" for a in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(np.ascontiguousarray(table, dtype="<f4").tobytes())


def read_ply(path) -> tuple[list[str], np.ndarray]:
    """Minimal reader for the files export_ply writes (tests / round trips)."""
    with open(path, "rb") as f:
        names, n = [], 0
        while True:
            line = f.readline().decode("ascii").strip()
            if line.startswith("element vertex"):
                n = int(line.split()[-1])
            elif line.startswith("property float"):
                names.append(line.split()[-1])
            elif line == "end_header":
                break
        data = np.frombuffer(f.read(), dtype="<f4").reshape(n, len(names))
    return names, data
