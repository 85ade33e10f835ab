"""The C-ABI is language-neutral: examples/c_abi_demo.c is a plain-C client (gcc, no Python / C++ / HIP headers).
Without a GPU it must build, link against every symbol it uses and fail loudly; on the GPU its checksums must
equal the same calls made through the Python binding."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "full_waveform_inversion_amd")


def build(tmp_path):
    exe = str(tmp_path / "c_abi_demo")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_abi_demo.c"), "-o", exe, "-L" + LIBDIR, "-lfwi_hip",
                           "-Wl,-rpath," + LIBDIR, "-lm"])
    return exe


def test_c_client_builds_and_fails_loudly_without_a_gpu(tmp_path):
    from full_waveform_inversion_amd import _lib
    exe = build(tmp_path)
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present: covered by the gpu test")
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 1 and "no HIP device" in p.stderr and "no CPU fallback" in p.stderr


@pytest.mark.gpu
def test_c_client_matches_python_binding(gpu, tmp_path):
    from full_waveform_inversion_amd import Engine
    exe = build(tmp_path)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    out = dict(line.split() for line in p.stdout.strip().splitlines())
    # the same shot through the Python binding
    nz, ny, nx, nt, nrec = 40, 36, 44, 60, 5
    i = np.arange(nz * ny * nx, dtype=np.uint64)
    x = ((i * np.uint64(2654435761)) % np.uint64(1000)).astype(np.float32)   # size_t arithmetic in the C client
    c = (np.float32(2000.0) + np.float32(500.0) * x / np.float32(1000.0)).astype(np.float32).reshape(nz, ny, nx)
    a = np.pi * 25.0 * (np.arange(nt) * 1.0e-3 - 0.04)
    wav = ((1.0 - 2.0 * a * a) * np.exp(-a * a)).astype(np.float32)
    rec = [[8, 6 + 5 * r, 7 + 6 * r] for r in range(nrec)]
    with Engine((nz, ny, nx), 10.0, 1.0e-3, nt, order=8, npml=6, sigma_max=400.0) as e:
        d = e.forward(c, ([[20, 18, 22]], wav), rec, save=True)
        adj = e.adjoint((np.float32(0.5) * d).astype(np.float32))
        g = e.gradient()
        assert out["kernel"] == e.kernel_name

    def checksum(x):
        x = np.asarray(x, np.float32).ravel()
        return float(np.sum(x.astype(np.float64) * ((np.arange(x.size) % 7) + 1)))

    for key, val in (("seis", d), ("adj", adj), ("grad", g)):
        ref = checksum(val)
        assert abs(float(out[key]) - ref) <= 1e-6 * max(abs(ref), 1e-30), (key, out[key], ref)
