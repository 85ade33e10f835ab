"""Generates tests/golden/*.npz from the NumPy oracle (oracle/fwi_oracle.py).

These are NOT reference vectors: the reference has no wave-propagation path
(SURVEY.md s.0), so nothing in /root/reference can produce them.  They freeze
the build-defined oracle's outputs on small seeded inputs so that (a) the
oracle's definition cannot drift unnoticed and (b) the GPU parity tests have
committed input/output pairs that travel to the GPU box.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import fwi_oracle as fo  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

CASES = {
    "g2d_o8": dict(shape=(48, 56), order=8, npml=8, nt=120, seed=11),
    "g2d_o2": dict(shape=(40, 36), order=2, npml=0, nt=80, seed=12),
    "g3d_o8": dict(shape=(24, 20, 28), order=8, npml=5, nt=60, seed=13),
    "g3d_o4": dict(shape=(18, 22, 20), order=4, npml=4, nt=50, seed=14),
}


def make(name, shape, order, npml, nt, seed):
    rng = np.random.default_rng(seed)
    nd = len(shape)
    c = 1500.0 + 1500.0 * rng.random(shape)
    # smooth a little so the fields are resolved, keep it heterogeneous
    for a in range(nd):
        c = 0.5 * c + 0.25 * (np.roll(c, 1, a) + np.roll(c, -1, a))
    h = 10.0
    dt = 0.7 * fo.cfl_dt(c.max(), h, nd, order)
    nsrc, nrec = 2, 9
    src = np.stack([rng.integers(npml, s - npml, nsrc) for s in shape], 1).astype(np.int32)
    rec = np.stack([rng.integers(0, s, nrec) for s in shape], 1).astype(np.int32)
    rec[0] = src[0]  # a receiver on top of a source
    w = np.stack([fo.ricker(nt, dt, 18.0), 0.5 * fo.ricker(nt, dt, 12.0, t0=0.05)], 1)
    p = fo.Propagator(c, h, dt, order, npml)
    seis = p.forward(src, w, rec)
    residual = seis * (1.0 + 0.3 * rng.standard_normal(seis.shape)) + 1e-3 * np.abs(seis).max() * \
        rng.standard_normal(seis.shape)
    adj_src = p.adjoint(residual)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), c=c, h=h, dt=dt, order=order, npml=npml,
        sigma_max=p.sigma_max, src_idx=src, rec_idx=rec, wavelet=w, seis=seis, residual=residual,
        adj_src=adj_src, grad_c=p.gradient("velocity"), grad_m=p.gradient("slowness2"))
    print(name, shape, "seis", seis.shape, "|seis|", np.linalg.norm(seis))


if __name__ == "__main__":
    for k, v in CASES.items():
        make(k, **v)
