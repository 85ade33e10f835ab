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
    c2 = (2000.0 + 300.0 * rng.random((40, 44))).astype(np.float32)
    # the round-2 options take their turns too: CPML, increment form (3-D stream / 2-D fused), bf16 store, off-grid
    # points spread on the device, the two-step 3-D kernel
    options = [dict(), dict(abc="cpml", pml_alpha_max=20.0), dict(update_form="increment"), dict(store_dtype="bf16"),
               dict(abc="cpml", update_form="increment")]
    for i in range(n):
        kw = options[i % len(options)]
        plain = not kw
        os.environ["FWI_STREAM_PAIR"] = "1" if i % 4 == 0 else "0"
        with Engine(shape, 10.0, 1e-3, 12, npml=3, sigma_max=200.0, ckpt_interval=(4 if plain and i % 3 == 0 else 0),
                    image_stride=(3 if plain and i % 3 == 1 else 1), **kw) as e:
            dd = e.forward(c, ([[12, 10, 14]], w), [[3, 4, 5], [20, 15, 22]], save=True)
            e.adjoint(dd)
            e.gradient()
            e.forward(None, ([[12, 10, 14]], w), [[3, 4, 5]], save=False)
            dp = e.forward_at(None, ([[11.3, 9.7, 13.2]], w), [[3.5, 4.25, 5.0], [20.0, 15.5, 22.75]], save=True)
            e.misfit_l2(0.9 * dp)
            e.adjoint(None)
            e.vec_create(2)
        if i % 5 == 0:
            with Engine((40, 44), 10.0, 1e-3, 12, npml=4, sigma_max=200.0,
                        **{k: v for k, v in kw.items() if k != "store_dtype"}) as e:
                dd = e.forward(c2, ([[20, 22]], w), [[5, 6], [30, 40]], save=True)
                e.adjoint(dd)
                e.gradient()
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
