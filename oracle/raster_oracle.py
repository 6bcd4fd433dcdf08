"""ctypes front-end of oracle/raster_oracle.c (CPU restatement of the rasterizer).

TEST INFRASTRUCTURE ONLY -- see the header of raster_oracle.c.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
PARITY UNPINNED (no reference golden vectors exist for this boundary).

Follows the call contract of /root/reference/src/model/decoder/cuda_splatting.py:100-127.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class FsoParams(C.Structure):
    _fields_ = [
        ("N", C.c_int), ("M", C.c_int), ("H", C.c_int), ("W", C.c_int), ("sh_degree", C.c_int),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float),
        ("bg", C.c_float * 3), ("view", C.c_float * 16), ("proj", C.c_float * 16),
        ("campos", C.c_float * 3),
    ]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libraster_oracle.so")
    src = os.path.join(_HERE, "raster_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libraster_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.fso_preprocess.restype = C.c_long
        _LIB.fso_exp_public.restype = C.c_float
        _LIB.fso_exp_public.argtypes = [C.c_float]
        _LIB.fso_set_exp_mode.argtypes = [C.c_int]
        _LIB.fso_set_exp_mode.restype = None
    return _LIB


def set_exp_mode(libm: bool) -> None:
    """Blend-loop exp: False = the arithmetic contract shared with the HIP kernels (default); True = libm expf
    (sensitivity studies only -- see raster_oracle.c:fso_set_exp_mode)."""
    lib().fso_set_exp_mode(1 if libm else 0)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def make_params(N, M, H, W, sh_degree, tanfovx, tanfovy, bg, view, proj, campos) -> FsoParams:
    P = FsoParams()
    P.N, P.M, P.H, P.W, P.sh_degree = int(N), int(M), int(H), int(W), int(sh_degree)
    P.tanfovx, P.tanfovy = float(tanfovx), float(tanfovy)
    P.bg[:] = [float(x) for x in np.asarray(bg, dtype=np.float32).reshape(3)]
    P.view[:] = [float(x) for x in np.asarray(view, dtype=np.float32).reshape(16)]
    P.proj[:] = [float(x) for x in np.asarray(proj, dtype=np.float32).reshape(16)]
    P.campos[:] = [float(x) for x in np.asarray(campos, dtype=np.float32).reshape(3)]
    return P


def forward(H, W, tanfovx, tanfovy, bg, viewmatrix, projmatrix, sh_degree, campos,
            means3D, cov3D, opacities, shs=None, colors_precomp=None, threads: int | None = None):
    """Returns a dict with every intermediate of the forward pipeline (numpy arrays)."""
    L = lib()
    if threads is not None:
        os.environ["OMP_NUM_THREADS"] = str(threads)
    means3D = _f32(means3D).reshape(-1, 3)
    N = means3D.shape[0]
    cov3D = _f32(cov3D).reshape(N, 6)
    opacities = _f32(opacities).reshape(N)
    assert (shs is None) != (colors_precomp is None)
    if shs is not None:
        shs = _f32(shs).reshape(N, -1, 3)
        M = shs.shape[1]
        assert (sh_degree + 1) ** 2 <= M and sh_degree <= 3
    else:
        colors_precomp = _f32(colors_precomp).reshape(N, 3)
        M = 0
    P = make_params(N, M, H, W, sh_degree, tanfovx, tanfovy, bg, viewmatrix, projmatrix, campos)
    depths = np.zeros(N, np.float32)
    radii = np.zeros(N, np.int32)
    means2D = np.zeros((N, 2), np.float32)
    conic_opacity = np.zeros((N, 4), np.float32)
    rgb = np.zeros((N, 3), np.float32)
    clamped = np.zeros((N, 3), np.uint8)
    rect = np.zeros((N, 4), np.int32)
    tiles_touched = np.zeros(N, np.uint32)
    I = L.fso_preprocess(C.byref(P), _p(means3D), _p(cov3D), _p(shs), _p(colors_precomp), _p(opacities),
                         _p(depths), _p(radii), _p(means2D), _p(conic_opacity), _p(rgb), _p(clamped),
                         _p(rect), _p(tiles_touched))
    gx, gy = (W + 15) // 16, (H + 15) // 16
    ranges = np.zeros((gx * gy, 2), np.uint32)
    point_list = np.zeros(max(int(I), 1), np.uint32)
    L.fso_bin(C.byref(P), _p(depths), _p(radii), _p(rect), _p(ranges), _p(point_list), C.c_long(I))
    point_list = point_list[: int(I)]
    color = np.zeros((3, H, W), np.float32)
    depth = np.zeros((H, W), np.float32)
    alpha = np.zeros((H, W), np.float32)
    final_T = np.zeros((H, W), np.float32)
    n_contrib = np.zeros((H, W), np.int32)
    pl = point_list if I > 0 else np.zeros(1, np.uint32)
    L.fso_render(C.byref(P), _p(ranges), _p(pl), _p(means2D), _p(conic_opacity), _p(rgb), _p(depths),
                 _p(color), _p(depth), _p(alpha), _p(final_T), _p(n_contrib))
    return dict(P=P, N=N, M=M, H=H, W=W, num_rendered=int(I), means3D=means3D, cov3D=cov3D, shs=shs,
                colors_precomp=colors_precomp, opacities=opacities, depths=depths, radii=radii,
                means2D=means2D, conic_opacity=conic_opacity, rgb=rgb, clamped=clamped, rect=rect,
                tiles_touched=tiles_touched, ranges=ranges, point_list=point_list, color=color,
                depth=depth, alpha=alpha, final_T=final_T, n_contrib=n_contrib)


def backward(st: dict, dL_dcolor, dL_ddepth=None):
    """Gradients w.r.t. means3D [N,3], cov3D [N,6], shs [N,M,3] | colors_precomp [N,3],
    opacities [N], and the screen-space means2D grad [N,2]."""
    L = lib()
    P, N = st["P"], st["N"]
    dL_dcolor = _f32(dL_dcolor).reshape(3, st["H"], st["W"])
    dL_ddepth = None if dL_ddepth is None else _f32(dL_ddepth).reshape(st["H"], st["W"])
    g_mean2D = np.zeros((N, 2), np.float64)
    g_conic = np.zeros((N, 3), np.float64)
    g_opac = np.zeros(N, np.float64)
    g_rgb = np.zeros((N, 3), np.float64)
    g_z = np.zeros(N, np.float64)
    pl = st["point_list"] if st["num_rendered"] > 0 else np.zeros(1, np.uint32)
    L.fso_render_backward(C.byref(P), _p(st["ranges"]), _p(pl), _p(st["means2D"]), _p(st["conic_opacity"]),
                          _p(st["rgb"]), _p(st["depths"]), _p(st["final_T"]), _p(st["n_contrib"]),
                          _p(dL_dcolor), _p(dL_ddepth), _p(g_mean2D), _p(g_conic), _p(g_opac), _p(g_rgb),
                          _p(g_z))
    f = lambda a: np.ascontiguousarray(a.astype(np.float32))
    gm2, gc, go, gr, gz = f(g_mean2D), f(g_conic), f(g_opac), f(g_rgb), f(g_z)
    have_sh = st["shs"] is not None
    d_means3D = np.zeros((N, 3), np.float32)
    d_cov3D = np.zeros((N, 6), np.float32)
    d_shs = np.zeros((N, max(st["M"], 1), 3), np.float32)
    d_colors = np.zeros((N, 3), np.float32)
    d_opac = np.zeros(N, np.float32)
    L.fso_preprocess_backward(C.byref(P), _p(st["means3D"]), _p(st["cov3D"]), _p(st["shs"]),
                              C.c_int(1 if have_sh else 0), _p(st["radii"]), _p(st["clamped"]),
                              _p(gm2), _p(gc), _p(go), _p(gr), _p(gz), _p(d_means3D), _p(d_cov3D),
                              _p(d_shs), _p(d_colors), _p(d_opac))
    return dict(means3D=d_means3D, cov3D=d_cov3D, shs=d_shs if have_sh else None,
                colors_precomp=None if have_sh else d_colors, opacities=d_opac, means2D=gm2,
                conic=gc, rgb=gr, z=gz)
