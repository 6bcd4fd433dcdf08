"""ctypes binding of libfreesplat_hip.so (C ABI: include/freesplat_amd.h).

The product path has NO fallback: if the HIP library is missing or a call fails this module
raises.  `build()` compiles the library in-tree with hipcc for gfx950.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
# FREESPLAT_LIB: an alternative build of the same library (A/B measurements of kernel variants inside one GPU session)
LIB_PATH = os.environ.get("FREESPLAT_LIB") or os.path.join(_PKG, "libfreesplat_hip.so")
_lib = None


class FreeSplatHipError(RuntimeError):
    pass


class RasterDims(C.Structure):
    _fields_ = [("N", C.c_int32), ("M", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("sh_degree", C.c_int32), ("tanfovx", C.c_float), ("tanfovy", C.c_float),
                ("flags", C.c_int32)]


ABI_VERSION = 6          # include/freesplat_amd.h FS_ABI_VERSION

RASTER_TILE_CULL = 1
RASTER_SH_FP16 = 2
RASTER_SH_CHANNEL_MAJOR = 4
RASTER_COV_FULL = 8
RASTER_FAST_EXP = 16
RASTER_NO_BACKWARD_STATE = 32


def build(force: bool = False) -> str:
    """Compile every HIP translation unit under csrc/ for gfx950 into libfreesplat_hip.so."""
    csrc = os.path.join(_PKG, "csrc")
    args = ["make", "-C", csrc, "-j8"]
    if force:
        subprocess.check_call(["make", "-C", csrc, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


_VP = C.c_void_p

# name -> (restype, argtypes); must list every symbol include/freesplat_amd.h declares
SIGNATURES = {
    "fs_version": (C.c_char_p, []),
    "fs_abi_version": (C.c_int, []),
    "fs_last_error": (C.c_char_p, []),
    "fs_profile_enable": (C.c_int, [C.c_int]),
    "fs_profile_collect": (C.c_int, [C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    "fs_profile_stage_name": (C.c_char_p, [C.c_int]),
    "fs_raster_buffer_sizes": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.POINTER(C.c_size_t)]),
    "fs_raster_forward": (C.c_int, [C.POINTER(RasterDims)] + [_VP] * 15 + [C.c_int64] + [_VP] * 6),
    "fs_raster_backward": (C.c_int, [C.POINTER(RasterDims)] + [_VP] * 24 + [C.c_int, _VP]),
    "fs_cost_volume_workspace_bytes": (C.c_size_t, [C.c_int32] * 5),
    "fs_cost_volume_forward": (C.c_int, [C.c_int32] * 6 + [_VP] * 6 + [C.c_int64] * 3 + [_VP] * 9),
    "fs_cost_volume_forward_layout": (C.c_int, [C.c_int32] * 6 + [_VP] * 6 + [C.c_int64] * 3 + [_VP] * 8 + [C.c_int32, _VP]),
    "fs_cost_volume_backward_workspace_bytes": (C.c_size_t, [C.c_int32] * 6),
    "fs_cost_volume_backward_workspace_bytes_for": (C.c_size_t, [C.c_int32] * 6 + [C.c_int64]),
    "fs_cost_volume_backward": (C.c_int, [C.c_int32] * 6 + [_VP] * 6 + [C.c_int64] * 3 + [_VP] * 16),
    "fs_cost_volume_depth_planes": (C.c_int, [C.c_int32] + [_VP] * 5),
    "fs_cost_volume_saved_bytes": (C.c_size_t, [C.c_int32] * 5),
    "fs_cost_volume_forward_train": (C.c_int, [C.c_int32] * 6 + [_VP] * 6 + [C.c_int64] * 3 + [_VP] * 10),
    "fs_cost_volume_backward_train": (C.c_int, [C.c_int32] * 6 + [_VP] * 6 + [C.c_int64] * 3 + [_VP] * 17),
    "fs_unproject_forward": (C.c_int, [C.c_int32] * 3 + [_VP] * 5),
    "fs_unproject_backward": (C.c_int, [C.c_int32] * 3 + [_VP] * 5),
    "fs_gaussian_head_forward": (C.c_int, [C.c_int64] + [_VP] * 4 + [C.c_int64, _VP, C.c_float, C.c_float] + [_VP] * 5),
    "fs_gaussian_head_backward": (C.c_int, [C.c_int64] + [_VP] * 4 + [C.c_int64, _VP, C.c_float, C.c_float] + [_VP] * 8),
    "fs_latents_pack_forward": (C.c_int, [C.c_int32, C.c_int64, C.c_int32] + [_VP] * 5),
    "fs_latents_pack_backward": (C.c_int, [C.c_int32, C.c_int64, C.c_int32] + [_VP] * 5),
    "fs_ptf_scratch_bytes": (C.c_size_t, [C.c_int32] * 3),
    "fs_ptf_match": (C.c_int, [C.c_int32] * 3 + [_VP] * 4 + [C.c_float] + [_VP] * 7),
    "fs_ptf_gru_inputs": (C.c_int, [C.c_int32] + [_VP] * 10),
    "fs_ptf_write_state": (C.c_int, [C.c_int32] * 3 + [_VP] * 24),
    "fs_ptf_gru_table_rows": (C.c_int32, []),
    "fs_ptf_gru_forward": (C.c_int, [C.c_int32] + [_VP] * 4),
    "fs_ptf_gru_table_t_rows": (C.c_int32, []),
    "fs_ptf_gru_stream_rows": (C.c_int32, []),
    "fs_ptf_gru_stream_layout": (C.c_int32, []),
    "fs_ptf_gru_table_layout": (C.c_int32, []),
    "fs_ptf_gru_stream_chunk_rows": (C.c_int32, []),
    "fs_ptf_gru_side_cols": (C.c_int32, []),
    "fs_ptf_gru_backward": (C.c_int, [C.c_int32] + [_VP] * 7),
    "fs_ptf_gru_backward_saved": (C.c_int, [C.c_int32] + [_VP] * 7),
    "fs_ptf_gru_act_cols": (C.c_int32, []),
    "fs_ptf_gru_stream_t_rows": (C.c_int32, []),
    "fs_ptf_gru_grad_floats": (C.c_int32, []),
    "fs_ptf_gru_weight_grads_bytes": (C.c_size_t, [C.c_int32]),
    "fs_ptf_gru_weight_grads": (C.c_int, [C.c_int32] + [_VP] * 5),
    "fs_raster_forward_views": (C.c_int, [C.POINTER(RasterDims), C.c_int32] + [_VP] * 15 + [C.POINTER(C.c_size_t), C.c_int64]
                                + [_VP] * 5 + [C.c_int32, C.POINTER(C.c_void_p), _VP]),
    "fs_raster_backward_views": (C.c_int, [C.POINTER(RasterDims), C.c_int32] + [_VP] * 15 + [C.POINTER(C.c_size_t)] + [_VP] * 9
                                 + [C.c_int32, C.c_int32, C.POINTER(C.c_void_p), _VP]),
    "fs_raster_backward_views_rows": (C.c_int, [C.POINTER(RasterDims), C.c_int32] + [_VP] * 15 + [C.POINTER(C.c_size_t)] + [_VP] * 9
                                      + [C.c_int32, C.c_int32, C.POINTER(C.c_void_p), _VP, C.c_int32, C.c_int32, C.c_int32]),
    "fs_ptf_fold_scratch_bytes": (C.c_size_t, [C.c_int32] * 3),
    "fs_ptf_fold_step": (C.c_int, [C.c_int32, _VP, C.c_int32, C.c_int32] + [_VP] * 14 + [C.c_float] + [_VP] * 10),
    "fs_ptf_fold_step_save": (C.c_int, [C.c_int32, _VP, C.c_int32, C.c_int32] + [_VP] * 14 + [C.c_float] + [_VP] * 13),
    "fs_ptf_fold_bytes": (C.c_size_t, [C.c_int32] * 3),
    "fs_ptf_fold": (C.c_int, [C.c_int32] * 3 + [_VP] * 8 + [C.c_float] + [_VP] * 2 + [C.POINTER(C.c_void_p)] * 2 + [_VP] * 2),
    "fs_ptf_cameras": (C.c_int, [C.c_int32] * 3 + [_VP] * 5),
    "fs_ptf_fold_step_lists": (C.c_int, [C.c_int32] * 3 + [_VP, C.POINTER(C.c_void_p)]),
    "fs_ptf_write_state_backward": (C.c_int, [C.c_int32] * 3 + [_VP] * 12 + [C.POINTER(C.c_void_p)] * 2 + [_VP] * 6),
    "fs_ptf_gru_inputs_backward": (C.c_int, [C.c_int32] + [_VP] * 14),
    "fs_frame_views": (C.c_int, [C.c_int32] + [_VP] * 4 + [C.c_int32] + [_VP] * 6),
    "fs_invert_4x4": (C.c_int, [C.c_int32, _VP, _VP, _VP]),
    "fs_depth_tail_forward": (C.c_int, [C.c_int32] * 4 + [_VP] * 2 + [C.c_int32] + [_VP] * 7),
    "fs_depth_tail_backward": (C.c_int, [C.c_int32] * 4 + [_VP] * 2 + [C.c_int32] + [_VP] * 13),
    "fs_raster_scratch_slots": (C.c_int, [C.c_int32, C.c_int32]),
    "fs_raster_tile_ranges": (_VP, [_VP, C.c_int32, C.c_int32]),
    "fs_raster_point_list": (_VP, [_VP, C.c_int32, C.c_int32]),
    "fs_raster_geom_records": (_VP, [_VP]),
    "fs_raster_final_T": (_VP, [_VP]),
    "fs_raster_n_contrib": (_VP, [_VP, C.c_int32, C.c_int32]),
}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FreeSplatHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). freesplat_amd has no CPU / eager fallback.")
        # torch first: it brings its own HIP runtime (libamdhip64 of its ROCm build), and the library must resolve its HIP
        # symbols to THAT copy -- loaded the other way round (this library first, as a bare `build(); smoke()` in one
        # process did) the system runtime is initialised beside torch's and every launch fails with "no ROCm-capable
        # device is detected"
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if L.fs_abi_version() != ABI_VERSION:
            raise FreeSplatHipError(f"{LIB_PATH} has ABI revision {L.fs_abi_version()}, this binding expects {ABI_VERSION} "
                                    "(include/freesplat_amd.h FS_ABI_VERSION): rebuild the library")
        _lib = L
    return _lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = lib().fs_last_error().decode(errors="replace")
        raise FreeSplatHipError(f"{what} failed with status {status} {msg}")


def ptr(t):
    """Device pointer of a torch tensor (or None).  The tensor must stay referenced until the library call that
    receives the pointer has returned (queued its launches): never write ptr(t.contiguous()) for more than one
    argument of a call -- a temporary freed between two conversions hands its block to the next one."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def profile_stage_names() -> list:
    L = lib()
    names = []
    while True:
        s = L.fs_profile_stage_name(len(names))
        if s is None:
            return names
        names.append(s.decode())


def profile_enable(on, stages=None) -> None:
    """on=False: off.  on=True: time every stage, or only the named `stages` (each timed launch costs two event
    records on the stream, so a throughput measurement should time as little as it needs)."""
    mask = 0
    if on:
        names = profile_stage_names()
        mask = -1 if stages is None else sum(1 << names.index(s) for s in stages)
    check(lib().fs_profile_enable(mask), "fs_profile_enable")


def profile_collect() -> dict:
    """{stage name: (total ms, launches)} of the launches recorded since the last collect."""
    L = lib()
    names = profile_stage_names()
    n = len(names)
    ms = (C.c_float * n)()
    cnt = (C.c_int32 * n)()
    check(L.fs_profile_collect(n, ms, cnt), "fs_profile_collect")
    return {names[i]: (float(ms[i]), int(cnt[i])) for i in range(n)}
