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


@pytest.fixture(autouse=True)
def _poisoned_device_memory(request):
    """Before every GPU test, fill a chunk of free device memory with NaN bit patterns and release it: the test's
    contexts are then likely to be handed pages full of NaNs rather than the zeros of a fresh process, so a buffer
    or halo that nobody initialised shows up as NaN in the result instead of passing by luck."""
    if request.node.get_closest_marker("gpu") is not None:
        import ctypes as C
        try:
            hip = C.CDLL("libamdhip64.so")
            ptr, n = C.c_void_p(), C.c_size_t(256 << 20)
            if hip.hipMalloc(C.byref(ptr), n) == 0:
                hip.hipMemset(ptr, 0xFF, n)
                hip.hipDeviceSynchronize()
                hip.hipFree(ptr)
        except OSError:
            pass
    yield
