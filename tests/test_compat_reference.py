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
    assert len(done) >= 8
    from freesplat_amd import encoder_forward
    assert enc.EncoderFreeSplat.forward is encoder_forward.encoder_forward
    # same state-dict keys as the reference modules they replace (checkpoint compatibility)
    ref_cv = importlib.reload(importlib.import_module("src.model.encoder.modules.cost_volume"))
    a = ref_cv.AVGFeatureVolumeManager(8, 8, num_depth_bins=4, mlp_channels=[202, 32, 32, 1], matching_dim_size=48)
    b = cost_volume.AVGFeatureVolumeManager(8, 8, num_depth_bins=4, mlp_channels=[202, 32, 32, 1], matching_dim_size=48)
    assert list(a.state_dict().keys()) == list(b.state_dict().keys())
    assert [tuple(v.shape) for v in a.state_dict().values()] == [tuple(v.shape) for v in b.state_dict().values()]
    from src.model.encoder.modules.networks import GRU as RefGRU
    assert list(RefGRU().state_dict().keys()) == list(ptf.GRU().state_dict().keys())


class _FakeBackbone:
    """Stand-in for timm's tf_efficientnetv2_s (pretrained weights: not available offline): a 5-level pyramid at strides
    2 .. 32 with that backbone's channel counts, one batch-norm layer included."""
    CH = [24, 48, 64, 160, 256]

    def __new__(cls):
        import types
        import torch
        from torch import nn

        class B(nn.Module):
            def __init__(self):
                super().__init__()
                self.convs = nn.ModuleList([nn.Conv2d(3, c, 3, padding=1) for c in cls.CH])
                self.bn = nn.BatchNorm2d(cls.CH[0])
                self.feature_info = types.SimpleNamespace(channels=lambda: list(cls.CH))

            def forward(self, x):
                ys = [c(nn.functional.avg_pool2d(x, 2 ** (i + 1))) for i, c in enumerate(self.convs)]
                return [self.bn(ys[0])] + ys[1:]
        return B()


def _same(a, b, path=""):
    import torch
    if torch.is_tensor(a):
        assert a.shape == b.shape and torch.equal(a, b), path
    elif isinstance(a, dict):
        assert sorted(a) == sorted(b), path
        for k in a:
            _same(a[k], b[k], f"{path}/{k}")
    elif isinstance(a, list):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _same(x, y, f"{path}[{i}]")
    elif hasattr(a, "__dataclass_fields__"):
        for k in a.__dataclass_fields__:
            _same(getattr(a, k), getattr(b, k), f"{path}.{k}")
    else:
        assert a == b, path


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("V,num_views,with_gt,b", [(3, 3, False, 1), (4, 3, True, 1), (2, 2, False, 2)])
def test_encoder_forward_equals_reference_forward(V, num_views, with_gt, b):
    """freesplat_amd.encoder_forward (bound as EncoderFreeSplat.forward by patch_reference) against the reference's own
    forward (encoder_freesplat.py:190-429) on the SAME, unpatched reference sub-modules (CPU): every entry of the two
    result dictionaries is identical -- the glue is the only thing that differs.  V > num_views exercises the
    pose-nearest source selection (:236-248)."""
    import types
    import torch
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_golden
    make_golden.install_shim()
    for k in [k for k in sys.modules if k.startswith("src.model.encoder.encoder_freesplat") or k.startswith("src.model.encoder.modules")
              or k.startswith("src.model.encoder.common")]:
        del sys.modules[k]                                     # (another test may have patched them)
    sys.modules["timm"].create_model = lambda *a, **k: _FakeBackbone()
    from src.model.encoder.encoder_freesplat import EncoderFreeSplat
    from src.model.encoder.common.gaussian_adapter import GaussianAdapterCfg
    import inputs
    from freesplat_amd.encoder_forward import encoder_forward
    h, w = 64, 96
    cfg = types.SimpleNamespace(name="freesplat", d_feature=64, num_surfaces=1, backbone=None, visualizer=None,
                                gaussian_adapter=GaussianAdapterCfg(gaussian_scale_min=0.5, gaussian_scale_max=15.0, sh_degree=2),
                                opacity_mapping=types.SimpleNamespace(initial=0.0, final=0.0, warm_up=1),
                                num_depth_candidates=16, num_views=num_views, image_H=h, image_W=w, log_planes=True)
    torch.manual_seed(0)
    enc = EncoderFreeSplat(cfg)
    assert type(enc).forward is not encoder_forward
    cams = [inputs.cameras(V, h // 4, w // 4, baseline=0.3, seed=3 + i) for i in range(b)]      # b scenes (b > 1: the per-scene loops)
    ctx = {"image": torch.rand(b, V, 3, h, w), "extrinsics": torch.stack([c[0] for c in cams]),
           "intrinsics": torch.stack([c[1] for c in cams]), "near": torch.full((b, V), 0.5), "far": torch.full((b, V), 15.0)}
    if with_gt:
        ctx["depth_s-1"] = 3.0 * torch.rand(b, V, 1, h, w)
        for s in range(4):
            ctx[f"depth_s{s}"] = 3.0 * torch.rand(b, V, 1, h >> (s + 1), w >> (s + 1))
    with torch.no_grad():
        ref = enc.forward(dict(ctx), 0)
        mine = encoder_forward(enc, dict(ctx), 0)
    assert ref["num_gaussians"] < V * h * w                       # something fused
    _same(ref, mine)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_launcher_dry_run_patches_the_reference():
    """python -m freesplat_amd.compat.run <module> --dry-run against the (stubbed) reference tree: install + patch, up to
    the point where the target module would run."""
    import subprocess
    root = os.path.dirname(HERE)
    code = ("import sys; sys.path[:0] = [%r, %r]; import make_golden; make_golden.install_shim(); "
            "sys.modules.pop('diff_gaussian_rasterization_depth', None); sys.modules.pop('src.model.decoder', None); "
            "sys.modules['src.dataset'].DatasetCfg = object; from freesplat_amd.compat import run; done = run.main(['src.main', '--dry-run']); "
            "import diff_gaussian_rasterization_depth as d, src.model.decoder as dec, src.model.encoder.encoder_freesplat as e; "
            "from freesplat_amd.decoder import DecoderSplattingCUDA; from freesplat_amd.encoder_forward import encoder_forward; "
            "assert d.GaussianRasterizer.__module__ == 'freesplat_amd.rasterizer'; "
            "assert dec.DECODERS['splatting_cuda'] is DecoderSplattingCUDA; assert e.EncoderFreeSplat.forward is encoder_forward; "
            "print('LAUNCHER_OK', len(done))") % (root, os.path.join(HERE, "golden"))
    p = subprocess.run([sys.executable, "-c", code], cwd=REF, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "LAUNCHER_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
