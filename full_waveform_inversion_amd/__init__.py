"""MI355X-native acoustic full-waveform-inversion hot path.

Drop-in entry points named by BASELINE.json's north_star -- ``forward(model,
src, rec)``, ``adjoint(residual)``, ``gradient()`` -- over hand-written gfx950
HIP kernels behind a ctypes C-ABI (include/fwi.h).  The reference repository
has no such path; see DESIGN.md for the provenance of the
scheme and of the oracle it is checked against.
"""
from __future__ import annotations

from ._lib import FwiError
from .engine import Engine, cfl_dt, default_sigma_max, ricker

__all__ = ["Engine", "FwiError", "cfl_dt", "default_sigma_max", "ricker", "configure", "forward",
           "adjoint", "gradient"]

_default = None


def configure(shape, h, dt, nt_max, **kw):
    """Create the module-level engine the three entry points below use.

    forward / adjoint / gradient together are an inversion's evaluation, so the engine is the one
    :func:`shots.inversion_engine` makes: fp32 in the low-round-off increment form (1e-5 end to end) unless
    ``update_form="standard"`` is passed -- the right choice for forward-only modelling, 4 B/update cheaper in 3-D."""
    global _default
    if _default is not None:
        _default.close()
    from .shots import inversion_engine
    _default = inversion_engine(shape, h, dt, nt_max, **kw)
    return _default


def _engine():
    if _default is None:
        raise FwiError(3, "call configure(shape, h, dt, nt_max, ...) first")
    return _default


def forward(model, src, rec, save=True):
    """Seismograms (nt, nrec) of ``model`` for ``src = (indices, wavelet)`` at receivers ``rec``."""
    return _engine().forward(model, src, rec, save=save)


def adjoint(residual, image=True):
    """Reverse-time propagation of ``residual``; accumulates the imaging condition."""
    return _engine().adjoint(residual, image=image)


def gradient(wrt="velocity"):
    """Gradient of 1/2 ||d - d_obs||^2 accumulated by the adjoint calls so far."""
    return _engine().gradient(wrt)
