"""Import-compatibility for an unmodified FreeSplat tree.

    import freesplat_amd.compat as compat
    compat.install()            # before `import src.main`: provides `diff_gaussian_rasterization_depth`
    compat.patch_reference()    # after `src` is importable: swaps the hot-path classes for the HIP ones

or, as ONE command from the root of the FreeSplat checkout (freesplat_amd on PYTHONPATH):

    python -m freesplat_amd.compat.run src.main +experiment=scannet/2views ...     (freesplat_amd/compat/run.py)
"""
import importlib
import sys


def install() -> None:
    """Register `diff_gaussian_rasterization_depth` (the module name FreeSplat imports at
    src/model/decoder/cuda_splatting.py:5) as an alias of freesplat_amd.rasterizer."""
    from . import diff_gaussian_rasterization_depth as m
    sys.modules.setdefault("diff_gaussian_rasterization_depth", m)


def patch_reference(decoder: bool = True) -> dict:
    """Rebind, inside the already importable reference package `src`, every name on the hot path to its
    MI355X implementation (same constructor / call signatures and state-dict keys, so configs and
    checkpoints are untouched):
      src.model.encoder.modules.cost_volume.AVGFeatureVolumeManager   (cost_volume.py:384)
      src.model.encoder.encoder_freesplat.{AVGFeatureVolumeManager, GaussianAdapter, GRU}
      src.model.encoder.encoder_freesplat.EncoderFreeSplat.fuse_gaussians       (:431)
      src.model.encoder.encoder_freesplat.EncoderFreeSplat.forward              (:190-429; encoder_forward.py: the
                                                                                 reference's sub-modules in the reference's
                                                                                 order, the repeat + gather glue of :216-288
                                                                                 replaced by direct indexing)
      src.model.encoder.modules.networks.DepthDecoder.forward                   (networks.py:108-154)
      src.model.decoder.DECODERS["splatting_cuda"]                              (decoder/__init__.py:5-13)
    Returns {dotted name: replacement} for logging."""
    from .. import cost_volume, depth_tail, encoder_forward, gaussian_adapter, ptf
    from ..decoder import DecoderSplattingCUDA
    done = {}
    cvm = importlib.import_module("src.model.encoder.modules.cost_volume")
    cvm.AVGFeatureVolumeManager = cost_volume.AVGFeatureVolumeManager
    done["src.model.encoder.modules.cost_volume.AVGFeatureVolumeManager"] = cost_volume.AVGFeatureVolumeManager
    enc = importlib.import_module("src.model.encoder.encoder_freesplat")
    enc.AVGFeatureVolumeManager = cost_volume.AVGFeatureVolumeManager
    enc.GaussianAdapter = gaussian_adapter.GaussianAdapter
    enc.GRU = ptf.GRU
    enc.EncoderFreeSplat.fuse_gaussians = ptf.fuse_gaussians
    enc.EncoderFreeSplat.forward = encoder_forward.encoder_forward
    done["src.model.encoder.encoder_freesplat.EncoderFreeSplat.forward"] = encoder_forward.encoder_forward
    for n in ("AVGFeatureVolumeManager", "GaussianAdapter", "GRU"):
        done[f"src.model.encoder.encoder_freesplat.{n}"] = getattr(enc, n)
    done["src.model.encoder.encoder_freesplat.EncoderFreeSplat.fuse_gaussians"] = ptf.fuse_gaussians
    net = importlib.import_module("src.model.encoder.modules.networks")
    net.DepthDecoder.forward = depth_tail.depth_decoder_forward
    done["src.model.encoder.modules.networks.DepthDecoder.forward"] = depth_tail.depth_decoder_forward
    if decoder:
        # same constructor (cfg, dataset_cfg) as the class it replaces, so get_decoder() is untouched.  A failure to
        # import the reference's decoder package is an error of the caller's environment and propagates.
        dec = importlib.import_module("src.model.decoder")
        dec.DECODERS["splatting_cuda"] = DecoderSplattingCUDA
        dec.DecoderSplattingCUDA = DecoderSplattingCUDA
        done['src.model.decoder.DECODERS["splatting_cuda"]'] = DecoderSplattingCUDA
    return done
