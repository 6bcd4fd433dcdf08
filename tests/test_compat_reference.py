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
    # the decoder package for real: its __init__ pulls `DatasetCfg` from the dataset package (stubbed like the rest
    # of the non-hot-path tree), then cuda_splatting.py imports OUR diff_gaussian_rasterization_depth
    import importlib, types
    sys.modules["src.dataset"].DatasetCfg = object
    sys.modules.pop("src.model.decoder", None)
    ref_dec = importlib.import_module("src.model.decoder")
    RefDecoder = ref_dec.DECODERS["splatting_cuda"]
    assert RefDecoder.__module__ == "src.model.decoder.decoder_splatting_cuda"
    done = compat.patch_reference(decoder=True)
    from freesplat_amd.decoder import DecoderSplattingCUDA
    assert ref_dec.DECODERS["splatting_cuda"] is DecoderSplattingCUDA
    cfg = types.SimpleNamespace(name="splatting_cuda")
    dataset_cfg = types.SimpleNamespace(background_color=[0.25, 0.5, 0.75])
    ours, theirs = ref_dec.get_decoder(cfg, dataset_cfg), RefDecoder(cfg, dataset_cfg)       # decoder/__init__.py:12-13
    assert isinstance(ours, DecoderSplattingCUDA) and ours.cfg is cfg and ours.dataset_cfg is dataset_cfg
    assert ours.background_color.tolist() == theirs.background_color.tolist() == [0.25, 0.5, 0.75]
    assert list(ours.state_dict().keys()) == list(theirs.state_dict().keys()) == []            # non-persistent buffer
    import inspect
    assert list(inspect.signature(ours.forward).parameters) == list(inspect.signature(theirs.forward).parameters)
    enc = sys.modules["src.model.encoder.encoder_freesplat"]
    assert enc.AVGFeatureVolumeManager is cost_volume.AVGFeatureVolumeManager
    assert enc.GaussianAdapter is gaussian_adapter.GaussianAdapter
    assert enc.EncoderFreeSplat.fuse_gaussians is ptf.fuse_gaussians
    from freesplat_amd import depth_tail
    assert sys.modules["src.model.encoder.modules.networks"].DepthDecoder.forward is depth_tail.depth_decoder_forward
    assert len(done) >= 7
    # same state-dict keys as the reference modules they replace (checkpoint compatibility)
    ref_cv = importlib.reload(importlib.import_module("src.model.encoder.modules.cost_volume"))
    a = ref_cv.AVGFeatureVolumeManager(8, 8, num_depth_bins=4, mlp_channels=[202, 32, 32, 1], matching_dim_size=48)
    b = cost_volume.AVGFeatureVolumeManager(8, 8, num_depth_bins=4, mlp_channels=[202, 32, 32, 1], matching_dim_size=48)
    assert list(a.state_dict().keys()) == list(b.state_dict().keys())
    assert [tuple(v.shape) for v in a.state_dict().values()] == [tuple(v.shape) for v in b.state_dict().values()]
    from src.model.encoder.modules.networks import GRU as RefGRU
    assert list(RefGRU().state_dict().keys()) == list(ptf.GRU().state_dict().keys())
