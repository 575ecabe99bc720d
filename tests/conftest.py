import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible (there is no CPU fallback)")
    return torch.device("cuda:0")


@pytest.fixture(params=["four_points_per_wave", "one_point_per_wave"])
def klt_kernel(request, monkeypatch):
    """Runs a tracker test once with each of the two 21x21 kernels (the library picks by the number of points; the
    environment variable, read at every call, overrides the limit)."""
    monkeypatch.setenv("VFE_KLT_WIDE_MAX", "0" if request.param == "four_points_per_wave" else str(2**31 - 1))
    return request.param
