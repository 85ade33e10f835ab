"""Lifecycle stress (a script, not collected by pytest): thousands of create / use / destroy cycles of the wave
engine context and of the Monte Carlo plan on one GPU.  Leaked streams, events or device buffers show up as a HIP
error or as free device memory shrinking between the first and the last hundred cycles."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from full_waveform_inversion_amd import Engine, source_inversion as si  # noqa: E402

hip = C.CDLL("libamdhip64.so")


def free_mb():
    fr, tot = C.c_size_t(0), C.c_size_t(0)
    hip.hipMemGetInfo(C.byref(fr), C.byref(tot))
    return fr.value / 1e6


def main(n=1500):
    rng = np.random.default_rng(0)
    shape = (24, 20, 28)
    c = (2000.0 + 300.0 * rng.random(shape)).astype(np.float32)
    w = rng.standard_normal(12).astype(np.float32)
    G = rng.standard_normal((4, 6, 32))
    d = np.einsum("kjt,j->kt", G, rng.standard_normal(6))
    marks = {}
    t0 = time.perf_counter()
    for i in range(n):
        with Engine(shape, 10.0, 1e-3, 12, npml=3, sigma_max=200.0, ckpt_interval=(4 if i % 3 == 0 else 0),
                    image_stride=(3 if i % 3 == 1 else 1)) as e:
            dd = e.forward(c, ([[12, 10, 14]], w), [[3, 4, 5], [20, 15, 22]], save=True)
            e.adjoint(dd)
            e.gradient()
            e.vec_create(2)
        with si.MonteCarloPlan(d, G, 512) as plan:
            plan.invert("full_mt", 300, seed=i)
        si.invert_on_device(d, G, 100, "DC", seed=i)
        if i in (100, n - 1):
            marks[i] = free_mb()
    print("%d cycles in %.1f s; free device memory after cycle 100: %.0f MB, after the last: %.0f MB"
          % (n, time.perf_counter() - t0, marks[100], marks[n - 1]))
    assert marks[n - 1] > marks[100] - 64.0, "device memory is leaking"
    print("ok")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1500)
