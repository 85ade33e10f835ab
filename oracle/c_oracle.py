"""ctypes binding of oracle/fwi_oracle.c -- TEST INFRASTRUCTURE ONLY (see fwi_oracle.py).

Same entry points as :class:`oracle.fwi_oracle.Propagator`, backed by the
OpenMP C restatement so parity tests can use grids the NumPy oracle would take
minutes on.  PARITY UNPINNED: the reference has no such path.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

from .fwi_oracle import _ravel_idx, default_sigma_max

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfwi_oracle.so")
_lib = None


def build(force=False):
    if os.environ.get("FWI_ORACLE_LIB"):  # another build of the same source (the sanitizer build, `make -C oracle asan`)
        return os.environ["FWI_ORACLE_LIB"]
    src = os.path.join(_HERE, "fwi_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libfwi_oracle.so"])
    return _SO


def default_threads():
    """Threads for the OpenMP loops: the CPUs this process may use, at most 16.

    The GPU boxes expose every host core but give a job a 16-CPU share; an
    uncapped OpenMP team oversubscribes that share and crawls.
    """
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))


def lib():
    global _lib
    if _lib is None:
        os.environ.setdefault("OMP_NUM_THREADS", str(default_threads()))
        _lib = ctypes.CDLL(build())
        f = _lib.fwi_oracle_propagate_abc
        f.restype = ctypes.c_int
        dp = ctypes.POINTER(ctypes.c_double)
        ip = ctypes.POINTER(ctypes.c_int64)
        f.argtypes = [ctypes.c_int] * 5 + [dp, ctypes.c_double, ctypes.c_double, ctypes.c_int,
                                           ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ip, dp, ctypes.c_double, ctypes.c_int, ip, ctypes.c_double,
                                           dp, ctypes.c_int, dp, dp, ctypes.c_int, ctypes.c_double]
    return _lib


def _dp(a):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))


class CPropagator:
    """fp64 forward/adjoint/gradient through the C oracle."""

    def __init__(self, c, h, dt, order=8, npml=0, sigma_max=None, threads=None, image_stride=1, abc="sponge",
                 pml_alpha_max=0.0):
        self.abc = {"sponge": 0, "cpml": 1}[abc]
        self.pml_alpha_max = float(pml_alpha_max)
        self.image_stride = max(1, int(image_stride))  # see Propagator: imaging every S-th step, weight S
        self.c = np.ascontiguousarray(c, np.float64)
        self.shape = self.c.shape
        self.ndim = self.c.ndim
        self.h, self.dt, self.order, self.npml = float(h), float(dt), int(order), int(npml)
        self.sigma_max = float(default_sigma_max(self.c.max(), h, npml) if sigma_max is None else sigma_max)
        self.q_store = None
        if threads:
            os.environ["OMP_NUM_THREADS"] = str(threads)

    def _run(self, reverse, inj, amp, inj_scale, rec, rec_scale, save_q, image):
        nz, nx = self.shape[0], self.shape[-1]
        ny = self.shape[1] if self.ndim == 3 else 1
        nt = amp.shape[0]
        out = np.zeros((nt, len(rec)))
        rc = lib().fwi_oracle_propagate_abc(
            self.ndim, nz, ny, nx, self.order, _dp(self.c), self.h, self.dt, self.npml,
            self.sigma_max, nt, int(reverse), len(inj), _ip(inj), _dp(amp), inj_scale, len(rec),
            _ip(rec), rec_scale, _dp(out), int(save_q), _dp(self.q_store), _dp(image), self.abc, self.pml_alpha_max)
        if rc:
            raise RuntimeError("fwi_oracle_propagate failed with code %d" % rc)
        return out

    def forward(self, src_idx, wavelet, rec_idx, save=True):
        w = np.ascontiguousarray(wavelet, np.float64)
        if w.ndim == 1:
            w = w[:, None]
        self.src_flat = np.ascontiguousarray(_ravel_idx(src_idx, self.shape))
        self.rec_flat = np.ascontiguousarray(_ravel_idx(rec_idx, self.shape))
        self.nt = w.shape[0]
        self.q_store = np.zeros((self.nt,) + self.shape) if save else None
        return self._run(False, self.src_flat, w, 1.0 / self.h ** self.ndim, self.rec_flat, 1.0,
                         save, None)

    def adjoint(self, residual, image=True):
        r = np.ascontiguousarray(residual, np.float64)
        self._img = np.zeros(self.shape) if image else None
        S = self.image_stride
        if image and S > 1:  # the C loop correlates every step: blank the terms the stride skips
            keep = np.zeros(self.nt, bool)
            keep[::S] = True
            self.q_store[~keep] = 0.0
        out = self._run(True, self.rec_flat, r, 1.0, self.src_flat, 1.0 / self.h ** self.ndim,
                        False, self._img)
        if image and S > 1:
            self._img *= S
        return out

    def gradient(self, wrt="velocity"):
        g_m = -self._img / self.dt ** 2
        return g_m if wrt == "slowness2" else g_m * (-2.0 / self.c ** 3)
