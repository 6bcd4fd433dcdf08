"""Drop-in wiring against the real reference tree (build container only: /root/reference is absent on the
GPU box, so this test skips there).  Uses the import shim of tests/golden/make_golden.py to make the
reference's encoder module importable, then freesplat_amd.compat.patch_reference()."""
import os
import sys

import pytest

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_patch_reference_rebinds_hot_path():
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_golden
    make_golden.install_shim()
    import freesplat_amd.compat as compat
    from freesplat_amd import cost_volume, gaussian_adapter, ptf
    sys.modules.pop("diff_gaussian_rasterization_depth", None)   # (the golden shim registers a stub)
    compat.install()
    import diff_gaussian_rasterization_depth as dgr
    assert dgr.GaussianRasterizer.__module__ == "freesplat_amd.rasterizer"
    assert dgr.GaussianRasterizationSettings._fields == ("image_height", "image_width", "tanfovx", "tanfovy", "bg",
                                                          "scale_modifier", "viewmatrix", "projmatrix", "sh_degree",
                                                          "campos", "prefiltered", "debug")      # cuda_splatting.py:100-113
    done = compat.patch_reference(decoder=False)
    enc = sys.modules["src.model.encoder.encoder_freesplat"]
    assert enc.AVGFeatureVolumeManager is cost_volume.AVGFeatureVolumeManager
    assert enc.GaussianAdapter is gaussian_adapter.GaussianAdapter
    assert enc.EncoderFreeSplat.fuse_gaussians is ptf.fuse_gaussians
    from freesplat_amd import depth_tail
    assert sys.modules["src.model.encoder.modules.networks"].DepthDecoder.forward is depth_tail.depth_decoder_forward
    assert len(done) >= 6
    # same state-dict keys as the reference modules they replace (checkpoint compatibility)
    import importlib
    ref_cv = importlib.reload(importlib.import_module("src.model.encoder.modules.cost_volume"))
    a = ref_cv.AVGFeatureVolumeManager(8, 8, num_depth_bins=4, mlp_channels=[202, 32, 32, 1], matching_dim_size=48)
    b = cost_volume.AVGFeatureVolumeManager(8, 8, num_depth_bins=4, mlp_channels=[202, 32, 32, 1], matching_dim_size=48)
    assert list(a.state_dict().keys()) == list(b.state_dict().keys())
    assert [tuple(v.shape) for v in a.state_dict().values()] == [tuple(v.shape) for v in b.state_dict().values()]
    from src.model.encoder.modules.networks import GRU as RefGRU
    assert list(RefGRU().state_dict().keys()) == list(ptf.GRU().state_dict().keys())
