import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "slow: full-size parity cases whose CPU oracle needs minutes (still part of `-m gpu`)")
    config.addinivalue_line("markers", "fast_exp: run the rasterizer with the opt-in hardware exp (FS_RASTER_FAST_EXP); "
                                       "every other test uses the default, bit-exact contract exp")


@pytest.fixture(autouse=True)
def _rasterizer_exp_mode(request):
    """The parity suite asserts BIT-exactness against the oracle, which only the contract exp can give: tests run in
    exact mode (the product default) unless marked `fast_exp` (those quantify the opt-in hardware exp).
    (Set directly, not through `monkeypatch`: tests that call monkeypatch.undo() must not flip the mode.)"""
    from freesplat_amd import rasterizer as R
    saved = R.FAST_EXP
    R.FAST_EXP = request.node.get_closest_marker("fast_exp") is not None
    yield
    R.FAST_EXP = saved


@pytest.fixture(scope="session")
def hip_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")
