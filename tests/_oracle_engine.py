"""Tests-only stand-in with the Engine interface, backed by the CPU oracle.

Lets the CPU multi-process tests drive the real shot-sharding / exchange /
optimiser host code without a GPU.  Never imported by the product package.
"""
import numpy as np

from oracle.c_oracle import CPropagator


class OracleEngine:
    def __init__(self, shape, h, dt, nt_max, order=8, npml=0, sigma_max=None, **_):
        self.shape, self.h, self.dt, self.order, self.npml = tuple(shape), h, dt, order, npml
        self.sigma_max = sigma_max
        self._p = None
        self._g = np.zeros(self.shape)

    def set_model(self, model):
        self._p = CPropagator(np.asarray(model, np.float64), self.h, self.dt, self.order, self.npml,
                              sigma_max=self.sigma_max)
        self.sigma_max = self._p.sigma_max

    def reset_gradient(self):
        self._g = np.zeros(self.shape)

    def forward(self, model, src, rec, save=True):
        if model is not None:
            self.set_model(model)
        return self._p.forward(src[0], src[1], rec, save=save)

    def adjoint(self, residual, image=True):
        a = self._p.adjoint(residual, image=image)
        if image:
            self._g += self._p.gradient("slowness2")
        return a

    def gradient(self, wrt="velocity"):
        return self._g if wrt == "slowness2" else self._g * (-2.0 / self._p.c ** 3)

    def gradient_add_from(self, other):
        self._g += other._g

    def close(self):
        pass

    def dot(self, a, b):
        return float(np.vdot(np.asarray(a, np.float64), np.asarray(b, np.float64)))
