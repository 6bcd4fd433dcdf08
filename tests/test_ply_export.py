"""N4: the PLY writer against a scipy restatement of the reference's ply_export.py:26-92 maths, plus a
write/read round trip."""
import numpy as np
import pytest
import torch

from freesplat_amd.ply_export import construct_list_of_attributes, export_ply, ply_attributes, read_ply


def _inputs(G=500, seed=0):
    g = torch.Generator().manual_seed(seed)
    E = torch.eye(4)
    a = 0.3
    E[:3, :3] = torch.tensor([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], dtype=torch.float32)
    means = torch.randn(G, 3, generator=g) * torch.tensor([2.0, 1.0, 3.0]) + torch.tensor([0.5, -1.0, 4.0])
    scales = torch.rand(G, 3, generator=g) * 0.05 + 0.001
    q = torch.randn(G, 4, generator=g); q = q / q.norm(dim=-1, keepdim=True)
    sh = torch.randn(G, 3, 9, generator=g)
    op = torch.randn(G, generator=g)
    return E, means, scales, q, sh, op


def test_attributes_match_scipy_restatement():
    R = pytest.importorskip("scipy.spatial.transform").Rotation
    E, means, scales, q, sh, op = _inputs()
    tab = ply_attributes(E, means, scales, q, sh, op)
    m = means - means.median(dim=0).values
    sf = m.abs().quantile(0.95, dim=0).max()
    m, s = m / sf, scales / sf
    rot = torch.tensor(R.from_rotvec([0, 0, -45], True).as_matrix(), dtype=torch.float32) @ \
        torch.tensor([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], dtype=torch.float32) @ E[:3, :3].inverse()
    m = (rot @ m.T).T
    rq = R.from_matrix(rot.numpy() @ R.from_quat(q.numpy()).as_matrix()).as_quat()
    wxyz = np.stack([rq[:, 3], rq[:, 0], rq[:, 1], rq[:, 2]], -1)
    np.testing.assert_allclose(tab[:, 0:3], m.numpy(), atol=1e-5)
    assert (tab[:, 3:6] == 0).all()
    np.testing.assert_allclose(tab[:, 6:9], sh[..., 0].numpy(), atol=0)
    np.testing.assert_allclose(tab[:, 9], op.numpy(), atol=0)
    np.testing.assert_allclose(tab[:, 10:13], s.log().numpy(), atol=1e-6)
    sign = np.sign((tab[:, 13:17] * wxyz).sum(-1, keepdims=True))       # q and -q are the same rotation
    np.testing.assert_allclose(tab[:, 13:17] * sign, wxyz, atol=1e-5)


def test_round_trip(tmp_path):
    E, means, scales, q, sh, op = _inputs(G=37, seed=3)
    path = tmp_path / "sub" / "scene.ply"
    export_ply(E, means, scales, q, sh, op, path)
    names, data = read_ply(path)
    assert names == construct_list_of_attributes(0) and len(names) == 17
    np.testing.assert_array_equal(data, ply_attributes(E, means, scales, q, sh, op))
    assert open(path, "rb").read(3) == b"ply"
