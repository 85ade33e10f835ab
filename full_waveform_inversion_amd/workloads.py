"""Deterministic synthetic workloads: BASELINE.json's five configs.

The reference ships no data (its inputs live on the author's laptop,
full_waveform_inversion.py:46) and has no grid/model at all, so every input is
generated here from fixed seeds.  ``scale`` shrinks grid and step count for
parity tests at sizes the CPU oracle finishes in seconds; ``scale=1`` is the
size BASELINE.json quotes.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from .engine import cfl_dt, ricker


@dataclass
class Workload:
    name: str
    c: np.ndarray            # velocity model (float64), (nz, nx) or (nz, ny, nx)
    h: float
    dt: float
    nt: int
    order: int
    npml: int
    f0: float
    src_idx: np.ndarray      # (nshots, ndim): one source per shot
    rec_idx: np.ndarray      # (nrec, ndim)
    c_init: np.ndarray | None = None  # starting model for inversion configs
    meta: dict = field(default_factory=dict)

    @property
    def shape(self):
        return self.c.shape

    @property
    def ndim(self):
        return self.c.ndim

    def wavelet(self, dtype=np.float32):
        return ricker(self.nt, self.dt, self.f0, dtype=dtype)

    @property
    def updates_per_shot(self):
        return int(np.prod(self.shape)) * self.nt


def _smooth(a, sigma):
    from scipy.ndimage import gaussian_filter
    return gaussian_filter(a, sigma, mode="nearest")


def cfg1(scale=1.0):
    """2-D constant velocity 256x256, 1 shot, 500 steps, O(2) (plumbing case)."""
    n = max(32, int(round(256 * scale)))
    nt = max(20, int(round(500 * scale)))
    c = np.full((n, n), 2000.0)
    h, order, npml = 10.0, 2, max(4, int(round(20 * scale)))
    dt = 0.8 * cfl_dt(2000.0, h, 2, order)
    src = np.array([[n // 2, n // 2]])
    xs = np.linspace(npml, n - npml - 1, min(64, n - 2 * npml)).astype(int)
    rec = np.stack([np.full_like(xs, max(npml // 2, 2)), xs], 1)
    return Workload("cfg1_2d_const_o2", c, h, dt, nt, order, npml, 10.0, src, rec)


def cfg2(scale=1.0, nshots=1):
    """2-D 1024x1024 four-layer model, 2000 steps, O(8) + absorbing border."""
    n = max(64, int(round(1024 * scale)))
    nt = max(40, int(round(2000 * scale)))
    c = np.empty((n, n))
    for i, v in enumerate([1500.0, 2000.0, 2500.0, 3000.0]):
        c[i * n // 4:(i + 1) * n // 4] = v
    h, order, npml = 5.0, 8, max(6, int(round(40 * scale)))
    dt = 0.8 * cfl_dt(3000.0, h, 2, order)
    zs = npml + 2
    if nshots == 1:
        sx = np.array([n // 2])
    else:
        sx = np.linspace(npml + 2, n - npml - 3, nshots).astype(int)
    src = np.stack([np.full_like(sx, zs), sx], 1)
    xs = np.linspace(npml, n - npml - 1, min(256, n - 2 * npml)).astype(int)
    rec = np.stack([np.full_like(xs, zs), xs], 1)
    return Workload("cfg2_2d_layered_o8", c, h, dt, nt, order, npml, 15.0, src, rec)


def cfg3(scale=1.0, nshots=32):
    """cfg2 grid, 32 shots, forward+adjoint+gradient at a smoothed starting model."""
    w = cfg2(scale, nshots=nshots)
    w.name = "cfg3_2d_32shots_gradient"
    w.c_init = _smooth(w.c, max(2.0, 20.0 * scale))
    return w


def cfg4(scale=1.0, npml=0):
    """3-D 256^3 constant velocity, 1 shot, 1000 steps, O(8): the HBM-roofline run.

    ``npml = 0`` (no absorbing border) is the headline setting; pass 16 for the
    damped variant.
    """
    n = max(24, int(round(256 * scale)))
    nt = max(20, int(round(1000 * scale)))
    c = np.full((n, n, n), 2000.0)
    h, order = 10.0, 8
    dt = 0.8 * cfl_dt(2000.0, h, 3, order)
    src = np.array([[n // 2, n // 2, n // 2]])
    k = min(16, n // 2)
    ys = np.linspace(n // 4, 3 * n // 4, k).astype(int)
    yy, xx = np.meshgrid(ys, ys, indexing="ij")
    rec = np.stack([np.full(k * k, min(8, n // 4)), yy.ravel(), xx.ravel()], 1)
    return Workload("cfg4_3d_const_o8", c, h, dt, nt, order, npml, 10.0, src, rec)


def cfg5(scale=1.0, nshots=64):
    """3-D 256^3 smooth random ("Marmousi-style") model, 64 shots, L-BFGS target."""
    n = max(24, int(round(256 * scale)))
    nt = max(20, int(round(1000 * scale)))
    rng = np.random.default_rng(0)
    f = _smooth(rng.standard_normal((n, n, n)), max(1.5, 12.0 * scale))
    f = (f - f.min()) / (f.max() - f.min())
    depth = np.linspace(0.0, 1.0, n)[:, None, None]
    c = 1500.0 + 3000.0 * np.clip(0.6 * f + 0.4 * depth, 0.0, 1.0)
    h, order, npml = 10.0, 8, max(4, int(round(16 * scale)))
    dt = 0.8 * cfl_dt(float(c.max()), h, 3, order)
    k = max(1, int(round(np.sqrt(nshots))))
    ss = np.linspace(npml + 2, n - npml - 3, k).astype(int)
    yy, xx = np.meshgrid(ss, ss, indexing="ij")
    src = np.stack([np.full(k * k, npml + 2), yy.ravel(), xx.ravel()], 1)[:nshots]
    kr = min(16, n // 2)
    rs = np.linspace(npml, n - npml - 1, kr).astype(int)
    ry, rx = np.meshgrid(rs, rs, indexing="ij")
    rec = np.stack([np.full(kr * kr, npml + 2), ry.ravel(), rx.ravel()], 1)
    w = Workload("cfg5_3d_random_o8", c, h, dt, nt, order, npml, 10.0, src, rec)
    w.c_init = _smooth(c, max(2.0, 30.0 * scale))
    return w


CONFIGS = {"cfg1": cfg1, "cfg2": cfg2, "cfg3": cfg3, "cfg4": cfg4, "cfg5": cfg5}
