"""Misfit layer (SURVEY s.8f-4): adjoint sources against finite differences, consistency with the
pinned restatement of the reference's similarity measures, and the result-file layout."""
import os
import pickle
import sys

import numpy as np
import pytest

from full_waveform_inversion_amd import io as fio, objectives as ob, shots as sh, workloads
from oracle import mc_oracle as mo

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _oracle_engine import OracleEngine  # noqa: E402


def _pair(seed=0, nt=120, k=5):
    rng = np.random.default_rng(seed)
    t = np.arange(nt)[:, None]
    o = np.sin(0.11 * t + rng.random((1, k)) * 3) * np.exp(-((t - 60) / 30.0) ** 2) + 0.1 * rng.standard_normal((nt, k))
    s = o * (0.7 + 0.6 * rng.random((1, k))) + 0.2 * rng.standard_normal((nt, k))
    return s, o


@pytest.mark.parametrize("name,kw", [("l2", {}), ("VR", {"per_trace": True}), ("VR", {"per_trace": False}),
                                     ("CC", {"per_trace": True}), ("CC", {"per_trace": False}),
                                     ("CC-shift", {"per_trace": True}), ("CC-shift", {"per_trace": False}),
                                     ("gau", {"per_trace": True}), ("gau", {"per_trace": False})])
def test_adjoint_source_is_the_derivative(name, kw):
    s, o = _pair()
    f = ob.OBJECTIVES[name]
    J, a = f(s, o, **kw)
    ds = np.random.default_rng(1).standard_normal(s.shape)
    eps = 1e-6
    fd = (f(s + eps * ds, o, **kw)[0] - f(s - eps * ds, o, **kw)[0]) / (2 * eps)
    assert abs(fd - np.sum(a * ds)) <= 1e-7 * max(abs(fd), 1e-3)


def test_misfits_are_one_minus_the_reference_similarities():
    """Against the pinned restatement of the reference's metrics (unclamped range)."""
    s, o = _pair(3)
    real, synth = o.T, s.T  # the reference's (k, t) layout
    for allat in (False, True):
        assert abs(ob.variance_reduction(s, o, per_trace=not allat)[0] -
                   (1 - mo.compare_synth_to_real_waveforms(real, synth, "VR", False, allat))) < 1e-12
        assert abs(ob.correlation(s, o, per_trace=not allat)[0] -
                   (1 - mo.compare_synth_to_real_waveforms(real, synth, "PCC", False, allat))) < 1e-12
        assert abs(ob.correlation(s, o, per_trace=not allat)[0] -
                   (1 - mo.compare_synth_to_real_waveforms(real, synth, "CC", False, allat))) < 1e-12
        assert abs(ob.correlation_shift(s, o, per_trace=not allat)[0] -
                   (1 - mo.compare_synth_to_real_waveforms(real, synth, "CC-shift", False, allat))) < 1e-12
    # gau: the reference's working branch is all-at-once (its per-trace branch returns 0, Appendix A-5);
    # the misfit is -ln of that similarity.  Scaled so that exp() does not underflow.
    s2 = o + 0.002 * (s - o)
    gau = mo.compare_synth_to_real_waveforms(o.T, s2.T, "gau", False, True)
    assert 0.0 < gau < 1.0 and abs(np.exp(-ob.gaussian(s2, o, per_trace=False)[0]) - gau) < 1e-12 * gau


@pytest.mark.parametrize("name", ["VR", "CC"])
def test_gradient_of_alternative_objective_through_the_shot_loop(name):
    """d/dc of the correlation / VR misfit via the adjoint source == finite differences (oracle engine)."""
    w = workloads.cfg3(0.0625, nshots=2)
    wav = w.wavelet(np.float64)
    shots = [sh.Shot(w.src_idx[i:i + 1], wav, w.rec_idx) for i in range(2)]
    e = OracleEngine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml)
    sh.model_data(e, w.c, shots)
    f = ob.OBJECTIVES[name]
    c0 = 0.97 * w.c_init
    J, g = sh.misfit_and_gradient(e, c0, shots, objective=f)
    dc = np.random.default_rng(2).standard_normal(c0.shape) * (c0 > 0)
    eps = 1e-2
    Jp = sh.misfit_and_gradient(e, c0 + eps * dc, shots, objective=f)[0]
    Jm = sh.misfit_and_gradient(e, c0 - eps * dc, shots, objective=f)[0]
    fd = (Jp - Jm) / (2 * eps)
    assert abs(fd - np.sum(g * dc)) < 2e-5 * abs(fd)


def test_io_round_trips_and_reference_result_layout(tmp_path):
    w = workloads.cfg2(0.0625)
    p = str(tmp_path / "m.npz")
    fio.save_model(p, w.c, w.h, note="x")
    c, h, meta = fio.load_model(p)
    assert np.array_equal(c, w.c) and h == w.h and "note" in meta
    shots = [sh.Shot(w.src_idx, w.wavelet(), w.rec_idx, np.ones((w.nt, len(w.rec_idx)), np.float32)),
             sh.Shot(w.src_idx, w.wavelet(), w.rec_idx)]
    fio.save_shots(str(tmp_path / "s.npz"), shots, w.dt)
    back, dt = fio.load_shots(str(tmp_path / "s.npz"))
    assert dt == w.dt and back[1].d_obs is None and np.array_equal(back[0].d_obs, shots[0].d_obs)
    MTs, MTp = np.random.default_rng(0).random((6, 10)), np.r_[np.zeros(3), np.ones(7) / 7]
    f = fio.save_to_MTFIT_style_file(MTs, MTp, "20180214185538", "full_mt", str(tmp_path), MTp_absolute=MTp * 2)
    assert os.path.basename(f) == "20180214185538_FW_full_mt.pkl"
    d = pickle.load(open(f, "rb"))
    assert set(d) == {"MTs", "MTp", "uid", "stations", "MTp_absolute"}  # the reference's keys (:961-967)
    p2, m2 = fio.remove_zero_prob_results(MTp, MTs)
    assert len(p2) == 7 and m2.shape == (6, 7)
    f = fio.save_specific_waveforms_to_file(MTs, MTs * 2, list("abcdef"), "u", "DC", str(tmp_path))
    d = pickle.load(open(f, "rb"))
    assert set(d["a"]) == {"real_wf", "synth_wf"}
