"""`diff_gaussian_rasterization_depth` as FreeSplat imports it (src/model/decoder/cuda_splatting.py:5-8),
backed by the MI355X HIP rasterizer.  Put the parent directory (freesplat_amd/compat) on PYTHONPATH."""
from freesplat_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer"]
