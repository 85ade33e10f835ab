#!/usr/bin/env python3
"""A small 2-D acoustic FWI from start to finish on one MI355X (a few seconds):

    python examples/fwi_2d_demo.py [--scale 0.25] [--shots 8] [--iters 10] [--out /tmp/fwi_demo.npz]

Layered 1500-3000 m/s model (BASELINE configs[2] at reduced size), observed data synthesised on it, inversion from a
smoothed start with L-BFGS on device-resident vectors; every misfit / gradient evaluation is `forward(model, src, rec)`
+ `adjoint(residual)` + `gradient()` per shot through the C-ABI.  For more than one GPU launch tools/run_config.py
under `python -m torch.distributed.run --nproc-per-node N` (one rank per GPU, one RCCL all-reduce per gradient).
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from full_waveform_inversion_amd import Engine, default_sigma_max, io, shots as sh, workloads  # noqa: E402
from full_waveform_inversion_amd.lbfgs import lbfgs_device  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=0.25, help="1.0 = 1024 x 1024 x 2000 steps")
    ap.add_argument("--shots", type=int, default=8)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    w = workloads.cfg3(a.scale, nshots=a.shots)
    wav = w.wavelet()
    shots = [sh.Shot(w.src_idx[i:i + 1], wav, w.rec_idx) for i in range(len(w.src_idx))]
    sigma = default_sigma_max(float(w.c.max()), w.h, w.npml)

    def make():
        return Engine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml, sigma_max=sigma)

    t0 = time.perf_counter()
    with sh.EnginePool(make, 2) as pool:          # two contexts overlap the 2-D shots on the GPU
        e = pool.primary
        sh.model_data(pool, w.c.astype(np.float32), shots)                   # "observed" data
        x0 = w.c_init.astype(np.float32)
        c, f, log = lbfgs_device(e, lambda xs, gs: sh.misfit_and_gradient_device(pool, xs, gs, shots), x0,
                                 maxiter=a.iters, history=5, first_step=30.0, bounds=(1000.0, 5000.0))
        kernel = e.kernel_name
    el = time.perf_counter() - t0
    err0 = np.linalg.norm(w.c_init - w.c) / np.linalg.norm(w.c)
    err1 = np.linalg.norm(c - w.c) / np.linalg.norm(w.c)
    print("grid %s, %d shots x %d steps, kernel %s: %d misfit evaluations in %.2f s"
          % ("x".join(map(str, w.shape)), len(shots), w.nt, kernel, log[-1]["evals"], el))
    print("misfit %.4e -> %.4e; model error %.4f -> %.4f (relative L2)" % (log[0]["f"], f, err0, err1))
    if a.out:
        io.save_model(a.out, c, w.h, misfit=f, iterations=len(log) - 1)
        print("inverted model written to", a.out)
    return 0 if f < 0.5 * log[0]["f"] else 1


if __name__ == "__main__":
    sys.exit(main())
