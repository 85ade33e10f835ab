"""Data misfits for the shot loop, with their adjoint sources.

The engine back-propagates whatever "residual" it is given; an objective is a function
``(d_syn, d_obs) -> (J, adjoint_source)`` with ``adjoint_source = dJ/d d_syn`` of the same
shape ``(nt, nrec)`` (time along axis 0, one column per trace).  Besides plain least squares
these are the reference's similarity measures turned into misfits ``1 - similarity`` (without
the reference's clamping at 0, which has no gradient):

* ``variance_reduction``  <- variance_reduction(), full_waveform_inversion.py:512-520
* ``correlation``         <- pearson_correlation_comparison() :568-576, which is numerically what
                             cross_corr_comparison() :534-546 also computes

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


OBJECTIVES = {"l2": l2, "VR": variance_reduction, "CC": correlation, "PCC": correlation}
