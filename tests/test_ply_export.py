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


def test_attributes_match_reference_fixture(tmp_path):
    """tests/golden/ply_small.npz: the attribute table the REFERENCE's export_ply (ply_export.py:26-92) handed to
    plyfile for a seeded 400-Gaussian scene (make_golden.gen_ply).  Same names in the same order; same numbers
    (quaternions up to the q / -q sign, which scipy's from_matrix picks differently in places)."""
    import os
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ply_small.npz"))
    t = lambda k: torch.from_numpy(fx[k])
    args = (t("extrinsics"), t("means"), t("scales"), t("rotations"), t("harmonics"), t("opacities"))
    assert list(fx["names"]) == construct_list_of_attributes(0)
    tab, ref = ply_attributes(*args), fx["table"]
    np.testing.assert_allclose(tab[:, 0:3], ref[:, 0:3], atol=2e-6)          # normalised, rotated means
    np.testing.assert_array_equal(tab[:, 3:10], ref[:, 3:10])                # normals (0), DC band, opacity: copies
    np.testing.assert_allclose(tab[:, 10:13], ref[:, 10:13], atol=1e-6)      # log scales
    sign = np.sign((tab[:, 13:17] * ref[:, 13:17]).sum(-1, keepdims=True))
    np.testing.assert_allclose(tab[:, 13:17] * sign, ref[:, 13:17], atol=2e-6)
    assert (sign > 0).mean() > 0.3                                           # (not a degenerate all-flipped match)
    path = tmp_path / "scene.ply"
    export_ply(*args, path)
    names, data = read_ply(path)
    assert names == list(fx["names"])
    np.testing.assert_array_equal(data, tab)
    head = open(path, "rb").read(200).decode("ascii", "replace")
    assert head.startswith("ply\nformat binary_little_endian 1.0\nelement vertex 400\nproperty float x\n")
