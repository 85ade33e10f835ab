"""Small file I/O: models / shot gathers as .npz, inversion results in the reference's layout.

The reference reads whitespace text traces with np.loadtxt (full_waveform_inversion.py:75-113)
from a directory on its author's machine and writes pickled dicts (:955-971, :1022-1035).  The
result writers below keep that dict layout so the reference's plotting scripts can read them;
the NonLinLoc lookup that supplies `uid` / `stations` there (:872-946, needs obspy) is replaced by
plain arguments.
"""
from __future__ import annotations

import os
import pickle

import numpy as np


def save_model(path, c, h, **meta):
    """Velocity model (m/s) with its grid spacing."""
    np.savez_compressed(path, c=np.asarray(c), h=float(h), **meta)


def load_model(path):
    z = np.load(path)
    return z["c"], float(z["h"]), {k: z[k] for k in z.files if k not in ("c", "h")}


def save_shots(path, shots, dt):
    """A list of shots.Shot (observed data included when present)."""
    out = {"dt": float(dt), "nshots": len(shots)}
    for i, s in enumerate(shots):
        out["src_%d" % i] = np.asarray(s.src_idx)
        out["wav_%d" % i] = np.asarray(s.wavelet)
        out["rec_%d" % i] = np.asarray(s.rec_idx)
        if s.d_obs is not None:
            out["obs_%d" % i] = np.asarray(s.d_obs)
    np.savez_compressed(path, **out)


def load_shots(path):
    from .shots import Shot
    z = np.load(path)
    shots = [Shot(z["src_%d" % i], z["wav_%d" % i], z["rec_%d" % i],
                  z["obs_%d" % i] if "obs_%d" % i in z.files else None) for i in range(int(z["nshots"]))]
    return shots, float(z["dt"])


def save_to_MTFIT_style_file(MTs, MTp, uid, inversion_type, outdir, stations=(), MTp_absolute=()):
    """``<outdir>/<uid>_FW_<inversion_type>.pkl`` with keys MTs, MTp, uid, stations[, MTp_absolute]
    -- the dict of the reference's writer (:955-971)."""
    out = {"MTs": np.asarray(MTs), "MTp": np.asarray(MTp), "uid": uid, "stations": list(stations)}
    if len(MTp_absolute) > 0:
        out["MTp_absolute"] = np.asarray(MTp_absolute)
    os.makedirs(outdir, exist_ok=True)
    fname = os.path.join(outdir, "%s_FW_%s.pkl" % (uid, inversion_type))
    with open(fname, "wb") as f:
        pickle.dump(out, f, protocol=2)  # protocol 2: readable from the reference's Python 2
    return fname


def save_specific_waveforms_to_file(real_data_array, synth_data_array, data_labels, uid, inversion_type,
                                    outdir):
    """``<uid>_FW_<type>.wfs``: {label: {"real_wf", "synth_wf"}} per trace (:1022-1035)."""
    out = {lab: {"real_wf": np.asarray(real_data_array[i]), "synth_wf": np.asarray(synth_data_array[i])}
           for i, lab in enumerate(data_labels)}
    os.makedirs(outdir, exist_ok=True)
    fname = os.path.join(outdir, "%s_FW_%s.wfs" % (uid, inversion_type))
    with open(fname, "wb") as f:
        pickle.dump(out, f, protocol=2)
    return fname


def remove_zero_prob_results(MTp, MTs):
    """Drop samples with zero posterior (:948-953)."""
    keep = np.asarray(MTp) > 0.0
    return np.asarray(MTp)[keep], np.asarray(MTs)[:, keep]
