"""Import-compatibility for an unmodified FreeSplat tree."""
import sys


def install() -> None:
    """Register `diff_gaussian_rasterization_depth` (the module name FreeSplat imports at
    src/model/decoder/cuda_splatting.py:5) as an alias of freesplat_amd.rasterizer."""
    from . import diff_gaussian_rasterization_depth as m
    sys.modules.setdefault("diff_gaussian_rasterization_depth", m)
