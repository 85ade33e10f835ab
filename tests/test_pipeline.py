"""The host functions either side of the Monte Carlo loop against the REFERENCE's outputs
(tests/golden/ref_pipeline.npz, made by tests/golden/make_reference_golden.py): trace loading with
time shift / phase cut / unit scalings (full_waveform_inversion.py:75-113, :168-197), the least-squares
estimate (:242-251), and -- on the GPU -- the best-sample synthetic (:974-1020) and the whole run()."""
import os
import pickle

import numpy as np
import pytest

from full_waveform_inversion_amd import io, samplers, source_inversion as si

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_pipeline.npz"))


def write_trace_files(datadir, real, mt, sf):
    names = ([], [], [])
    for i in range(real.shape[0]):
        for lst, stem, arr in ((names[0], "real_%02d.txt", real[i]), (names[1], "gf_mt_%02d.txt", mt[i].T),
                               (names[2], "gf_sf_%02d.txt", sf[i].T)):
            lst.append(stem % i)
            np.savetxt(os.path.join(datadir, stem % i), arr, fmt="%.18e")
    return names


@pytest.fixture(scope="module")
def traces(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("traces"))
    return (d,) + write_trace_files(d, Z["real"], Z["mt"], Z["sf"])


def test_load_input_data_vs_reference(traces):
    d, rn, mn, sn = traces
    shift_mt, shift_sf = list(Z["shift_mt"]), list(Z["shift_sf"])
    for key, got in (("plain", io.load_input_data(d, rn, mn)),
                     ("shift", io.load_input_data(d, rn, mn, shift_mt)),
                     ("shiftkeep", io.load_input_data(d, rn, mn, shift_mt,
                                                      set_pre_time_shift_values_to_zero_switch=False)),
                     ("cut", io.load_input_data(d, rn, sn, shift_sf, list(Z["cut_start"]), int(Z["cut_len"])))):
        assert np.array_equal(got[0], Z[key + "_real"]) and np.array_equal(got[1], Z[key + "_green"]), key
    assert Z["cut_green"].shape == (5, 3, 60) and Z["plain_green"].shape == (5, 6, 120)


@pytest.mark.parametrize("typ", samplers.INVERSION_TYPES)
def test_overall_data_and_least_squares_vs_reference(traces, typ):
    d, rn, mn, sn = traces
    real, green = io.get_overall_real_and_green_func_data(
        d, rn, mn, sn, typ, list(Z["shift_mt"]), list(Z["shift_sf"]), list(Z["cut_start"]), int(Z["cut_len"]))
    assert np.array_equal(real, Z["overall_real_" + typ])
    assert green.shape == (5, samplers.NUM_COMPONENTS[typ], 60)
    assert np.allclose(green, Z["overall_green_" + typ], rtol=1e-15, atol=0)
    M = si.perform_inversion(real, green)
    ref = Z["lsq_" + typ]
    assert M.shape == ref.shape == (samplers.NUM_COMPONENTS[typ], 1)
    assert np.allclose(M, ref, rtol=1e-9, atol=1e-12 * np.abs(ref).max())


def test_unknown_type_raises(traces):
    d, rn, mn, sn = traces
    with pytest.raises(ValueError):
        io.get_overall_real_and_green_func_data(d, rn, mn, sn, "nope")


@pytest.mark.gpu
@pytest.mark.parametrize("typ", samplers.INVERSION_TYPES)
def test_most_likely_synthetic_vs_reference(gpu, typ):
    synth = si.get_synth_forward_model_most_likely_result(Z["ml_MTs_" + typ], Z["ml_MTp_" + typ],
                                                          Z["overall_green_" + typ], typ)
    ref = Z["ml_synth_" + typ]
    assert synth.shape == ref.shape and np.allclose(synth, ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())


@pytest.mark.gpu
@pytest.mark.parametrize("typ", ["full_mt", "DC_single_force_no_coupling"])
def test_run_writes_the_reference_layout(gpu, traces, tmp_path, typ):
    """run(): least-squares + Monte Carlo on real files; with data synthesised from a known source the
    least-squares estimate recovers it and the best Monte Carlo sample fits better than a random one."""
    d, rn, mn, sn = traces
    real, green = io.get_overall_real_and_green_func_data(d, rn, mn, sn, typ)
    n = samplers.NUM_COMPONENTS[typ]
    M_true = np.random.default_rng(4).standard_normal(n)
    synth = np.einsum("kjt,j->kt", green, M_true)
    dd = str(tmp_path / "data")
    os.makedirs(dd)
    rn2, mn2, sn2 = write_trace_files(dd, synth, Z["mt"], Z["sf"])
    out = str(tmp_path / "out")
    labels = ["ST%02d, Z" % i for i in range(len(rn2))]
    MTs, MTp, MTp_abs = si.run(dd, out, rn2, mn2, sn2, labels, typ, False, False, 20000, "VR", uid="ev1",
                               stations=["ST00"], return_absolute_similarity_values_switch=True, seed=3)
    coupled = typ in samplers.COUPLED_TYPES
    assert MTs.shape == (n + coupled, 20000) and abs(MTp.sum() - 1) < 1e-9 and MTp_abs.shape == (20000,)
    with open(os.path.join(out, "least_squares_result", "ev1_FW_%s.pkl" % typ), "rb") as f:
        lsq = pickle.load(f)
    assert np.allclose(lsq["MTs"][:, 0], M_true, rtol=1e-6) and lsq["MTp"][0] > 0.999999
    with open(os.path.join(out, "ev1_FW_%s.pkl" % typ), "rb") as f:
        res = pickle.load(f)
    assert sorted(res) == ["MTp", "MTp_absolute", "MTs", "stations", "uid"] and np.array_equal(res["MTs"], MTs)
    with open(os.path.join(out, "ev1_FW_%s.wfs" % typ), "rb") as f:
        wfs = pickle.load(f)
    assert sorted(wfs) == sorted(labels) and np.array_equal(wfs[labels[2]]["real_wf"], synth[2])
    # samples are scaled to the least-squares amplitude (:1172, :741)
    amp = np.linalg.norm(M_true)
    if not coupled:
        assert np.allclose(np.linalg.norm(MTs, axis=0), amp, rtol=1e-9)
    best = int(np.argmax(MTp))
    assert MTp_abs[best] > np.median(MTp_abs)


HYP = """NLLOC "x" "LOCATED" "Location completed."
GEOGRAPHIC  OT 2014 06 29  18 42   10.123456  Lat -75.1 Long -84.2 Depth 2.0
PHASE ID Ins Cmp On Pha  FM Date     HrMn   Sec     Err  ErrMag    Coda      Amp       Per  >   TTpred    Res       Weight    StaLoc(X  Y         Z)        SDist    SAzim  RAz  RDip RQual    Tcorr
ST01   ?    ?    ? P      ? 20140629 1842   10.5000 GAU  2.00e-02 -1.00e+00 -1.00e+00 -1.00e+00 >     0.3770  0.0000    1.0000    1.0000    2.0000    0.0000    0.8000 110.00 110.0  95.0  9     0.0000
ST01   ?    ?    ? S      ? 20140629 1842   10.9000 GAU  2.00e-02 -1.00e+00 -1.00e+00 -1.00e+00 >     0.7770  0.0000    1.0000    1.0000    2.0000    0.0000    0.8000 110.00 111.0  96.0  9     0.0000
ST03   ?    ?    ? S      ? 20140629 1842   11.2000 GAU  2.00e-02 -1.00e+00 -1.00e+00 -1.00e+00 >     0.9770  0.0000    1.0000    5.0000    2.0000    0.0000    1.8000  10.00  11.0  70.0  9     0.0000
ST02   ?    ?    ? P      ? 20140629 1842   10.6000 GAU  2.00e-02 -1.00e+00 -1.00e+00 -1.00e+00 >     0.4770  0.0000    1.0000    3.0000    1.0000    0.0000    0.9000 250.00 250.5 100.0  9     0.0000
END_PHASE
END_NLLOC
"""


def test_nonlinloc_hyp_reader(tmp_path):
    """Event uid and MTFIT-style station entries from a NonLinLoc .hyp file (:872-946, restated without obspy / grep /
    awk; not pinned: the reference's function needs obspy): origin time -> uid with microseconds, one entry per
    station with a P phase (azimuth = field 22, take-off = 180 - field 24), polarity 0."""
    f = tmp_path / "event.hyp"
    f.write_text(HYP)
    uid, stations = io.get_event_uid_and_station_data_MTFIT_FORMAT_from_nonlinloc_hyp_file(str(f))
    assert uid == "20140629184210123456"
    assert [s[0][0] for s in stations] == ["ST01", "ST02"]          # ST03 has no P phase
    assert [float(s[1][0, 0]) for s in stations] == [110.0, 250.0]
    assert [float(s[2][0, 0]) for s in stations] == [85.0, 80.0]
    assert all(s[1].shape == s[2].shape == s[3].shape == (1, 1) and int(s[3][0, 0]) == 0 for s in stations)
    (tmp_path / "bad.hyp").write_text("NLLOC\nEND_NLLOC\n")
    with pytest.raises(ValueError):
        io.get_event_uid_and_station_data_MTFIT_FORMAT_from_nonlinloc_hyp_file(str(tmp_path / "bad.hyp"))
