"""Data misfits for the shot loop, with their adjoint sources.

The engine back-propagates whatever "residual" it is given; an objective is a function
``(d_syn, d_obs) -> (J, adjoint_source)`` with ``adjoint_source = dJ/d d_syn`` of the same
shape ``(nt, nrec)`` (time along axis 0, one column per trace).  Besides plain least squares
these are the reference's similarity measures turned into misfits ``1 - similarity`` (without
the reference's clamping at 0, which has no gradient):

* ``variance_reduction``  <- variance_reduction(), full_waveform_inversion.py:512-520
* ``correlation``         <- pearson_correlation_comparison() :568-576, which is numerically what
                             cross_corr_comparison() :534-546 also computes

* ``correlation_shift``  <- cross_corr_comparison_shift_allowed() :548-566.  The reference rolls BOTH
                             signals by the same number of samples (:561-562), so every "shift" scores the
                             same and the function is the zero-lag correlation of the two signals after 4x
                             linear upsampling; that is what is differentiated here.
* ``gaussian``           <- gaussian_comparison() :578-582, ``gau = exp(-sum (d - s)^2 / (2 sigma^2))`` with
                             the noise level ``sigma = mean |d[-60:-10]|`` taken from the data.  As a misfit
                             this is the negative log of it, ``J = -ln gau`` (noise-weighted least squares):
                             ``1 - gau`` itself underflows to a constant 1 with zero gradient as soon as the
                             fit is a few sigma off.  ``exp(-J)`` is the reference's similarity.

``per_trace=True`` averages the per-trace values with equal weights like the reference's
compare_all_waveforms_simultaneously=False branch (:682); ``False`` treats all traces as one
flattened signal (:609-621).
"""
from __future__ import annotations

import numpy as np


def l2(d_syn, d_obs):
    """J = 1/2 ||d_syn - d_obs||^2."""
    r = np.asarray(d_syn, np.float64) - np.asarray(d_obs, np.float64)
    # np.sum, not a BLAS dot: a multi-threaded BLAS leaves spinning worker threads behind that
    # steal the CPU from the thread enqueueing the next shot's kernel launches (measured: 2-D
    # shots 30 -> 100 ms on a 16-CPU share)
    return 0.5 * float(np.sum(r * r)), r


def variance_reduction(d_syn, d_obs, per_trace=True):
    """J = 1 - VR = sum (d_obs - d_syn)^2 / sum d_obs^2  (mean over traces if per_trace)."""
    s, o = np.asarray(d_syn, np.float64), np.asarray(d_obs, np.float64)
    r = s - o
    if per_trace:
        den = np.sum(o * o, axis=0)
        k = s.shape[1]
        return float(np.sum(np.sum(r * r, axis=0) / den) / k), 2.0 * r / den / k
    den = float(np.sum(o * o))
    return float(np.sum(r * r)) / den, 2.0 * r / den


def correlation(d_syn, d_obs, per_trace=True):
    """J = 1 - r, r = Pearson / zero-lag normalised correlation (mean over traces if per_trace)."""
    s, o = np.asarray(d_syn, np.float64), np.asarray(d_obs, np.float64)
    ax = 0 if per_trace else None
    sc = s - np.mean(s, axis=ax, keepdims=True)
    oc = o - np.mean(o, axis=ax, keepdims=True)
    ns = np.sqrt(np.sum(sc * sc, axis=ax, keepdims=True))
    no = np.sqrt(np.sum(oc * oc, axis=ax, keepdims=True))
    r = np.sum(sc * oc, axis=ax, keepdims=True) / (ns * no)
    dr = oc / (ns * no) - r * sc / (ns * ns)  # already mean-free, so centring needs no extra term
    if per_trace:
        k = s.shape[1]
        return float(np.sum(1.0 - r) / k), -dr / k
    return float(1.0 - r.reshape(-1)[0]), -dr


def _upsample(x, factor=4):
    """np.interp(np.arange(0, n, 1 / factor), np.arange(n), x) along axis 0 (:553-554), as explicit
    gather weights so that the transpose below is exact.  Positions past the last sample repeat it."""
    n = x.shape[0]
    pos = np.arange(0.0, n, 1.0 / factor)
    i0 = np.minimum(np.floor(pos).astype(np.int64), n - 1)
    i1 = np.minimum(i0 + 1, n - 1)
    w1 = (pos - i0)[:, None]
    return (1.0 - w1) * x[i0] + w1 * x[i1], (i0, i1, w1)


def _upsample_T(y, taps, n):
    i0, i1, w1 = taps
    out = np.zeros((n,) + y.shape[1:])
    np.add.at(out, i0, (1.0 - w1) * y)
    np.add.at(out, i1, w1 * y)
    return out


def correlation_shift(d_syn, d_obs, per_trace=True):
    """J = 1 - CC-shift: the correlation misfit of the 4x linearly upsampled traces (see the module note)."""
    s, o = np.asarray(d_syn, np.float64), np.asarray(d_obs, np.float64)
    if not per_trace:  # the reference flattens all traces into one signal BEFORE upsampling (:609-621)
        shape = s.shape
        s, o = s.T.reshape(-1, 1), o.T.reshape(-1, 1)
    su, taps = _upsample(s)
    ou, _ = _upsample(o)
    J, a = correlation(su, ou, per_trace=True)
    a = _upsample_T(a, taps, s.shape[0])
    return J, (a if per_trace else a.reshape(shape[1], shape[0]).T)


def gaussian(d_syn, d_obs, per_trace=False):
    """J = -ln gau = sum (d_obs - d_syn)^2 / (2 sigma^2), sigma = mean |d_obs[-60:-10]| (:580-581).

    ``per_trace=False`` (default) is the reference's working branch: one sigma from the flattened data.
    ``per_trace=True`` gives every trace its own sigma and averages (the reference's per-trace branch assigns
    to the wrong variable and always returns 0; this is what it evidently meant)."""
    s, o = np.asarray(d_syn, np.float64), np.asarray(d_obs, np.float64)
    r = s - o
    if per_trace:
        sig = np.mean(np.abs(o[-60:-10]), axis=0)
        k = s.shape[1]
        return float(np.sum(np.sum(r * r, axis=0) / (2.0 * sig ** 2)) / k), r / sig ** 2 / k
    sig = float(np.mean(np.abs(o.T.reshape(-1)[-60:-10])))  # the flattened signal is trace after trace
    return float(np.sum(r * r)) / (2.0 * sig ** 2), r / sig ** 2


OBJECTIVES = {"l2": l2, "VR": variance_reduction, "CC": correlation, "PCC": correlation,
              "CC-shift": correlation_shift, "gau": gaussian}
