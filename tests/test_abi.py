"""CPU checks of the drop-in boundary: libfreesplat_hip.so loads without a GPU and exports every
symbol include/freesplat_amd.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "freesplat_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fs_[a-z0-9_A-Z]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    from freesplat_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert "fs_raster_forward" in names and "fs_raster_backward" in names
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/freesplat_amd.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in freesplat_amd/_lib.py"
    assert set(_lib.SIGNATURES) <= set(names)
    assert _lib.lib().fs_version().decode().endswith("gfx950")


def test_buffer_sizes_and_arg_validation():
    from freesplat_amd import _lib
    L = _lib.lib()
    out = (ctypes.c_size_t * 4)()
    assert L.fs_raster_buffer_sizes(1_000_000, 968, 1296, 8_000_000, out) == 0
    geom, binning, image, scratch = list(out)
    assert geom >= 1_000_000 * (48 + 8 + 1)
    assert binning >= 8_000_000 * 4 + (81 * 61 + 1) * 4
    assert image >= 968 * 1296 * 8
    assert scratch >= 8_000_000 * 8
    assert L.fs_raster_buffer_sizes(-1, 10, 10, 10, out) == -1
    assert L.fs_raster_buffer_sizes(10, 0, 10, 10, out) == -1
    # NULL dims -> FS_ERR_INVALID_ARG before anything touches a device
    args = [None] * 16 + [1] + [None] * 6
    assert L.fs_raster_forward(*args) == -1


def test_product_has_no_oracle_dependency():
    """The product package must never import/execute anything under oracle/."""
    pkg = os.path.join(ROOT, "freesplat_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "raster_oracle" not in txt and "oracle/" not in txt, f
