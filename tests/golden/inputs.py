"""Seeded synthetic input generators shared by tests/golden/make_golden.py (which feeds them to the
reference) and by the tests (which feed the same inputs to the oracle / HIP path).  Pure torch."""
import numpy as np
import torch


def cameras(V, h, w, baseline=0.25, seed=0):
    """c2w extrinsics [V,4,4] (small arc, slight rotation), normalised intrinsics [V,3,3] fx != fy."""
    g = torch.Generator().manual_seed(seed)
    E = torch.eye(4).repeat(V, 1, 1)
    for i in range(V):
        t = 0.0 if V == 1 else i / (V - 1) - 0.5
        ang = 0.08 * t
        R = torch.tensor([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], dtype=torch.float32)
        E[i, :3, :3] = R
        E[i, :3, 3] = torch.tensor([baseline * t, 0.02 * t, 0.01 * t]) + 0.003 * torch.randn(3, generator=g)
    K = torch.tensor([[0.9, 0, 0.49], [0, 1.2, 0.51], [0, 0, 1]]).repeat(V, 1, 1)
    return E, K



def cv_inputs(V, K, h4, w4, C, seed, behind=False, oblique=0.0):
    """The 8 kwargs of cost_volume.forward as EncoderFreeSplat.forward prepares them
    (encoder_freesplat.py:216-288) for b=1, V views, each with its K nearest = all-other views.
    `oblique` (radians): the last view is turned about its y axis by that much -- from ~1 rad on, the horizon of the depth
    planes crosses its image (rays parallel to the planes: projections through infinity, tiles partly behind the camera)."""
    torch.manual_seed(seed)
    E, Kn = cameras(V, h4, w4, seed=seed)
    if oblique:
        R = torch.tensor([[np.cos(oblique), 0, np.sin(oblique)], [0, 1, 0], [-np.sin(oblique), 0, np.cos(oblique)]], dtype=torch.float32)
        E[-1, :3, :3] = E[-1, :3, :3] @ R
    if behind:  # one source looks away / sits in front of the points: exercises the z>0 mask + zero padding
        E[-1, :3, 3] += torch.tensor([0.0, 0.0, 1.2])
        E[-1, :3, :3] = E[-1, :3, :3] @ torch.tensor([[-1.0, 0, 0], [0, 1, 0], [0, 0, -1.0]])
    feats = torch.randn(V, C, h4, w4)
    Kp = Kn.clone()
    Kp[:, 0] *= w4
    Kp[:, 1] *= h4
    K44 = torch.eye(4).repeat(V, 1, 1)
    K44[:, :3, :3] = Kp
    invK44 = torch.linalg.inv(K44)
    src_idx = [[j for j in range(V) if j != i][:K] for i in range(V)]
    src_feats = torch.stack([feats[idx] for idx in src_idx])                     # [V,K,C,h,w]
    src_cam_T_cur = torch.stack([torch.linalg.inv(E[idx]) @ E[i] for i, idx in enumerate(src_idx)])  # [V,K,4,4]
    cur_cam_T_src = torch.linalg.inv(src_cam_T_cur)
    src_Ks = torch.stack([K44[idx] for idx in src_idx])
    return dict(cur_feats=feats, src_feats=src_feats, src_extrinsics=src_cam_T_cur, src_poses=cur_cam_T_src,
                src_Ks=src_Ks, cur_invK=invK44, min_depth=torch.tensor(0.5).view(1, 1, 1, 1),
                max_depth=torch.tensor(15.0).view(1, 1, 1, 1))



def ptf_inputs(V, h, w, seed, tie=False):
    torch.manual_seed(seed)
    E, Kn = cameras(V, h, w, baseline=0.3, seed=seed)
    depths = 2.0 + 0.02 * torch.randn(V, 1, h, w)
    if tie:
        depths[:] = 2.0
        E[:] = torch.eye(4)          # identical cameras + constant depth: exact z ties / duplicate pixels
        E[1, 0, 3] = 0.0
    lat = torch.randn(1, V, h * w, 64)
    dens = torch.rand(1, V, h * w, 1, 1)
    wts = torch.rand(1, V, h * w, 1, 1)
    return E, Kn, depths, lat, dens, wts


