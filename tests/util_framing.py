"""Torch restatement of the reference's camera framing -- TEST INFRASTRUCTURE (the checker), not the product.

  get_fov                 /root/reference/src/geometry/projection.py:233-247
  get_projection_matrix   /root/reference/src/model/decoder/cuda_splatting.py:17-44
  _frame                  the per-view matrices exactly as cuda_splatting.py:64-87 chains them in fp32

The product frames all views of a call in one HIP launch (freesplat_amd.decoder.frame_views -> fs_frame_views,
csrc/framing.hip); these functions run on CPU tensors and are what tests/golden/framing.npz (generated from the reference
itself by make_golden.py) and the oracle-side view inputs are checked with / built from.
"""
import torch
from torch import Tensor

_EDGE_MID = torch.tensor([[0, 0.5, 1], [1, 0.5, 1], [0.5, 0, 1], [0.5, 1, 1]], dtype=torch.float32)


def get_fov(intrinsics: Tensor) -> Tensor:
    """[B,3,3] normalised intrinsics -> [B,2] (fov_x, fov_y): angle between the unit rays through the
    image-edge midpoints (projection.py:233-247)."""
    inv = torch.linalg.inv_ex(intrinsics).inverse  # same LU as .inverse(), without its host sync
    mids = _EDGE_MID.to(intrinsics.device)

    def ray(k):
        r = torch.einsum("bij,j->bi", inv, mids[k])
        return r / r.norm(dim=-1, keepdim=True)

    left, right = ray(0), ray(1)
    top, bottom = ray(2), ray(3)
    fov_x = (left * right).sum(dim=-1).acos()
    fov_y = (top * bottom).sum(dim=-1).acos()
    return torch.stack((fov_x, fov_y), dim=-1)


def get_projection_matrix(near: Tensor, far: Tensor, fov_x: Tensor, fov_y: Tensor) -> Tensor:
    """Symmetric frustum, x/y -> (-1,1), z -> (0,1), w = z (cuda_splatting.py:17-44)."""
    tan_x = (0.5 * fov_x).tan()
    tan_y = (0.5 * fov_y).tan()
    top = tan_y * near
    bottom = -top
    right = tan_x * near
    left = -right
    (b,) = near.shape
    P = torch.zeros((b, 4, 4), dtype=torch.float32, device=near.device)
    P[:, 0, 0] = 2 * near / (right - left)
    P[:, 1, 1] = 2 * near / (top - bottom)
    P[:, 0, 2] = (right + left) / (right - left)
    P[:, 1, 2] = (top + bottom) / (top - bottom)
    P[:, 3, 2] = 1
    P[:, 2, 2] = far / (far - near)
    P[:, 2, 3] = -(far * near) / (far - near)
    return P


def _frame(extrinsics, intrinsics, near, far, scale_invariant: bool):
    """Per-view matrices exactly as cuda_splatting.py:64-87 builds them."""
    scale = None
    if scale_invariant:
        scale = 1 / near
        extrinsics = extrinsics.clone()
        extrinsics[..., :3, 3] = extrinsics[..., :3, 3] * scale[:, None]
        near = near * scale
        far = far * scale
    fov_x, fov_y = get_fov(intrinsics).unbind(dim=-1)
    tan_fov_x = (0.5 * fov_x).tan()
    tan_fov_y = (0.5 * fov_y).tan()
    projection = get_projection_matrix(near, far, fov_x, fov_y).transpose(1, 2)
    view = torch.linalg.inv_ex(extrinsics).inverse.transpose(1, 2)
    full = view @ projection
    return extrinsics, scale, tan_fov_x, tan_fov_y, view.contiguous(), full.contiguous()
