"""Off-grid sources and receivers by multilinear interpolation, on top of the node-based engine.

The engine (like the scheme in DESIGN.md s.2) injects and samples at grid nodes.  A point at
fractional grid coordinates ``(z, x)`` / ``(z, y, x)`` (in cells, node ``i`` at coordinate ``i``)
is spread over the ``2**D`` surrounding nodes with the usual bi- / trilinear weights:

    sampling   d_r   = sum_j w_rj u[node_rj]              ``gather``
    injection  f     = sum_j w_sj delta(node_sj) s(t)     ``scatter``

``scatter`` and ``gather`` are exact transposes of each other, so a shot with interpolated points
keeps the adjoint identity of the node-based operators and its gradient stays the exact gradient
of the interpolated forward map.  Nodes outside the grid are dropped (the field is zero there).

    S, R = Spread(src_xyz, shape), Spread(rec_xyz, shape)
    d = R.gather(engine.forward(model, (S.idx, S.scatter(wavelet)), R.idx, save=True))
    adj = S.gather(engine.adjoint(R.scatter(residual)))
    g = engine.gradient()

No reference counterpart; nearest-node points remain the default everywhere.
"""
from __future__ import annotations

import itertools

import numpy as np


class Spread:
    """Multilinear spreading of ``n`` points at fractional grid coordinates onto grid nodes."""

    def __init__(self, coords, shape):
        coords = np.atleast_2d(np.asarray(coords, dtype=np.float64))
        shape = tuple(int(s) for s in shape)
        if coords.ndim != 2 or coords.shape[1] != len(shape):
            raise ValueError("coords must be (n, %d) for a grid of shape %s" % (len(shape), shape))
        lim = np.array(shape, dtype=np.float64) - 1.0
        if np.any(coords < 0.0) or np.any(coords > lim):
            raise ValueError("coordinates must lie inside the grid [0, n - 1] on every axis")
        base = np.minimum(np.floor(coords), lim - 1.0).clip(min=0.0)  # lower corner; last cell for x == n - 1
        frac = coords - base
        idx, wts, own = [], [], []
        for corner in itertools.product((0, 1), repeat=len(shape)):
            c = np.array(corner)
            w = np.prod(np.where(c == 1, frac, 1.0 - frac), axis=1)
            node = (base + c).astype(np.int64)
            ok = (w != 0.0) & np.all(node <= lim.astype(np.int64), axis=1)
            idx.append(node[ok])
            wts.append(w[ok])
            own.append(np.nonzero(ok)[0])
        self.n = coords.shape[0]
        order = np.argsort(np.concatenate(own), kind="stable")  # entries grouped by point (the device gather's CSR)
        self.idx = np.ascontiguousarray(np.concatenate(idx)[order], dtype=np.int32)   # (m, D) nodes
        self.weights = np.concatenate(wts)[order]                                      # (m,)
        self.owner = np.concatenate(own)[order]                                        # (m,) point of each node entry
        self.pt_start = np.concatenate(([0], np.cumsum(np.bincount(self.owner, minlength=self.n)))).astype(np.int32)

    def scatter(self, a):
        """Per-point time series ``(nt, n)`` (or ``(nt,)`` for one point) -> per-node series ``(nt, m)``."""
        a = np.asarray(a)
        if a.ndim == 1:
            a = a[:, None]
        if a.shape[1] != self.n:
            raise ValueError("expected (nt, %d)" % self.n)
        return np.ascontiguousarray(a[:, self.owner] * self.weights.astype(a.dtype))

    def gather(self, x):
        """Per-node time series ``(nt, m)`` -> per-point series ``(nt, n)``; the transpose of ``scatter``."""
        x = np.asarray(x)
        if x.ndim != 2 or x.shape[1] != len(self.owner):
            raise ValueError("expected (nt, %d)" % len(self.owner))
        out = np.zeros((x.shape[0], self.n), dtype=x.dtype)
        np.add.at(out, (slice(None), self.owner), x * self.weights.astype(x.dtype))
        return out
