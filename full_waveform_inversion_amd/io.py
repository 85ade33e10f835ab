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


def get_event_uid_and_station_data_MTFIT_FORMAT_from_nonlinloc_hyp_file(nlloc_hyp_filename):
    """Event uid and station list from a NonLinLoc ``.hyp`` file (:872-946), without obspy or shell tools.

    The reference greps the ``GEOGRAPHIC`` line (origin time = its fields 2-7: year month day hour minute seconds)
    and the lines between ``PHASE ID`` and ``END_PHASE`` (field 0 station, 4 phase, 22 azimuth, 24 ray take-off
    angle), and builds ``uid = strftime("%Y%m%d%H%M%S%f")`` and one MTFIT-style entry per station that has a P phase:
    ``[name (1,), azimuth (1, 1), 180 - take-off (1, 1), polarity 0 (1, 1)]``.  NOT pinned to the reference: its
    function needs obspy's ``UTCDateTime``, which is absent here (the only unpinned function of the source-inversion
    side); stations come in order of first appearance (the reference's order is Python 2's dict order).
    """
    geo, phases, inside = None, [], False
    with open(nlloc_hyp_filename) as f:
        for line in f:
            if "GEOGRAPHIC" in line and geo is None:
                geo = line.split()
            if "END_PHASE" in line:
                inside = False
            elif inside and line.strip():
                phases.append(line.split())
            elif "PHASE ID" in line:
                inside = True
    if geo is None or len(geo) < 8:
        raise ValueError("%s: no GEOGRAPHIC line with an origin time" % nlloc_hyp_filename)
    sec = float(geo[7])
    micro = int(round((sec - int(sec)) * 1e6))
    carry, micro = divmod(micro, 1000000)
    import datetime
    t0 = datetime.datetime(int(geo[2]), int(geo[3]), int(geo[4]), int(geo[5]), int(geo[6]), int(sec)) \
        + datetime.timedelta(seconds=carry, microseconds=micro)
    uid = t0.strftime("%Y%m%d%H%M%S%f")
    angles = {}
    for p in phases:
        if len(p) > 24 and p[4] == "P":
            angles[p[0]] = (float(p[22]), 180.0 - float(p[24]))  # (a later P line of a station overrides, as :927-929)
    stations = [[np.array([name], dtype=str), np.array([[azi]], dtype=float), np.array([[toa]], dtype=float),
                 np.array([[0]], dtype=int)] for name, (azi, toa) in angles.items()]
    return uid, stations


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


def load_input_data(datadir, real_data_fnames, green_func_fnames, manual_indices_time_shift=(),
                    cut_phase_start_vals=(), cut_phase_length=0, set_pre_time_shift_values_to_zero_switch=True):
    """The reference's trace loader (:75-113): one whitespace text file per trace for the data
    (``t`` values) and for the Green's functions (``t`` rows x ``n`` component columns).

    Returns ``real_data_array (k, t)`` and ``green_func_array (k, n, t)``.  Optional per-trace
    alignment: the Green's functions are rolled by ``manual_indices_time_shift[i]`` samples along
    time (samples rolled in from the end are zeroed unless the switch is off), then a window
    ``[start_i, start_i + cut_phase_length)`` is cut from both.
    """
    real = np.stack([np.loadtxt(os.path.join(datadir, f), dtype=float) for f in real_data_fnames])
    green = np.stack([np.transpose(np.loadtxt(os.path.join(datadir, f), dtype=float)) for f in green_func_fnames])
    if len(manual_indices_time_shift) > 0:
        rolled = np.zeros_like(green)  # traces beyond the list keep zeros, as in the reference (:95-99)
        for i, shift in enumerate(manual_indices_time_shift):
            rolled[i] = np.roll(green[i], shift, axis=1)
            if set_pre_time_shift_values_to_zero_switch:
                rolled[i, :, 0:shift] = 0.0
        green = rolled
    if len(cut_phase_start_vals) > 0:
        n = int(cut_phase_length)
        real = np.stack([real[i, int(s):int(s) + n] for i, s in enumerate(cut_phase_start_vals)])
        green = np.stack([green[i, :, int(s):int(s) + n] for i, s in enumerate(cut_phase_start_vals)])
    return real, green


def load_input_data_multiple_media(datadir, real_data_fnames, green_func_fnames, green_func_fnames_split_index,
                                   manual_indices_time_shift=(), cut_phase_start_vals=(), cut_phase_length=0,
                                   set_pre_time_shift_values_to_zero_switch=True):
    """The reference's two-media trace loader (:116-165): ``green_func_fnames`` holds the files of medium 1 followed,
    from ``green_func_fnames_split_index`` on, by the files of medium 2 for the same traces.

    Returns ``real_data_array (k, t)`` and ``green_func_array (k, n, t, 2)`` (last axis = medium); time shift and phase
    cut as in :func:`load_input_data`, applied to both media alike.
    """
    split = int(green_func_fnames_split_index)
    media = (list(green_func_fnames[:split]), list(green_func_fnames[split:]))
    if len(media[0]) != len(media[1]):
        raise ValueError("Green's-function file list does not split into two media of equal length at index %d "
                         "(%d / %d files)" % (split, len(media[0]), len(media[1])))  # (:123-125 print + exit)
    real = np.stack([np.loadtxt(os.path.join(datadir, f), dtype=float) for f in real_data_fnames])
    ntr = len(real_data_fnames)  # the reference loads one Green's-function file per DATA trace (:138)
    green = np.stack([np.stack([np.transpose(np.loadtxt(os.path.join(datadir, media[m][i]), dtype=float))
                                for i in range(ntr)]) for m in (0, 1)], axis=3)
    if len(media[0]) > ntr:  # further files are allocated but never read (:134): zeros
        green = np.concatenate((green, np.zeros((len(media[0]) - ntr,) + green.shape[1:])), axis=0)
    if len(manual_indices_time_shift) > 0:
        rolled = np.zeros_like(green)
        for i, shift in enumerate(manual_indices_time_shift):
            rolled[i] = np.roll(green[i], shift, axis=1)
            if set_pre_time_shift_values_to_zero_switch:
                rolled[i, :, 0:shift, :] = 0.0
        green = rolled
    if len(cut_phase_start_vals) > 0:
        n = int(cut_phase_length)
        cut_r = np.zeros((real.shape[0], n))
        cut_g = np.zeros((green.shape[0], green.shape[1], n, 2))
        for i in range(ntr):
            s0 = int(cut_phase_start_vals[i])
            cut_r[i] = real[i, s0:s0 + n]
            cut_g[i] = green[i, :, s0:s0 + n, :]
        real, green = cut_r, cut_g
    return real, green


def get_overall_real_and_green_func_data(datadir, real_data_fnames, MT_green_func_fnames,
                                         single_force_green_func_fnames, inversion_type,
                                         manual_indices_time_shift_MT=(), manual_indices_time_shift_SF=(),
                                         cut_phase_start_vals=(), cut_phase_length=0,
                                         set_pre_time_shift_values_to_zero_switch=True,
                                         invert_for_ratio_of_multiple_media_greens_func_switch=False,
                                         green_func_fnames_split_index=0):
    """Data and Green's functions for an inversion type (:168-197): moment-tensor files for the
    6-component types, single-force files for ``single_force``, both side by side (9 components)
    for the combined types; moment-tensor Green's functions are scaled by 1e3 (units relative to
    the single-force ones) and everything by 1e7 (SI), as the reference does.  With
    ``invert_for_ratio_of_multiple_media_greens_func_switch`` the file lists hold two media
    (:func:`load_input_data_multiple_media`) and the Green's functions come back ``(k, n, t, 2)``.
    """
    kw = dict(cut_phase_start_vals=cut_phase_start_vals, cut_phase_length=cut_phase_length,
              set_pre_time_shift_values_to_zero_switch=set_pre_time_shift_values_to_zero_switch)
    if invert_for_ratio_of_multiple_media_greens_func_switch:
        def load(datadir, real_fnames, green_fnames, shift, **kw2):
            return load_input_data_multiple_media(datadir, real_fnames, green_fnames, green_func_fnames_split_index,
                                                  shift, **kw2)
    else:
        load = load_input_data
    if inversion_type in ("full_mt", "DC", "DC_crack_couple"):
        real, green = load(datadir, real_data_fnames, MT_green_func_fnames, manual_indices_time_shift_MT, **kw)
        green = green * (10 ** 3)
    elif inversion_type == "single_force":
        real, green = load(datadir, real_data_fnames, single_force_green_func_fnames, manual_indices_time_shift_SF,
                           **kw)
    elif inversion_type in ("DC_single_force_couple", "DC_single_force_no_coupling",
                            "single_force_crack_no_coupling"):
        real, mt = load(datadir, real_data_fnames, MT_green_func_fnames, manual_indices_time_shift_MT, **kw)
        real, sf = load(datadir, real_data_fnames, single_force_green_func_fnames, manual_indices_time_shift_SF, **kw)
        green = np.hstack((mt * (10 ** 3), sf))
    else:
        raise ValueError("unknown inversion_type %r" % (inversion_type,))
    return real, green * (10 ** 7)
