import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_count():
    from full_waveform_inversion_amd import _lib
    return _lib.device_count()  # raises ImportError if the HIP library is not built: no fallback


@pytest.fixture(scope="session")
def gpu():
    """Fails (not skips) when the HIP library is missing; skips only when no device is visible."""
    n = _gpu_count()
    if n < 1:
        pytest.skip("no HIP device visible")
    return n
