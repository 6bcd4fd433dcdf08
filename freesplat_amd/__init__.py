"""freesplat_amd -- MI355X-native (gfx950) implementation of FreeSplat's data-parallel hot path.

Only what the path needs lives here: `csrc/` (hand-written HIP kernels + the C ABI declared in
include/freesplat_amd.h), and the host-side mirrors of the reference's operator interfaces:

  rasterizer.py   GaussianRasterizationSettings / GaussianRasterizer  (diff_gaussian_rasterization_depth)
  decoder.py      frame_views / render_cuda / render_views / DecoderSplattingCUDA (src/model/decoder/)
  compat/         importable `diff_gaussian_rasterization_depth` module for an unmodified reference tree

There is no CPU or eager fallback: the ops raise if libfreesplat_hip.so is missing.
"""
__version__ = "0.1.0"
