"""Pins oracle/mc_oracle.py to the reference: golden vectors made by the reference's own code
(tests/golden/make_reference_golden.py, lib2to3 in memory) for forward_model, all five
similarity metrics in all four dispatcher modes, the likelihood map and the posterior."""
import os
import warnings

import numpy as np
import pytest

from oracle import mc_oracle as mo

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["ref_mc_fullmt", "ref_mc_force", "ref_mc_dc"]
METRICS = ["VR", "CC", "PCC", "CC-shift", "gau"]


@pytest.mark.parametrize("name", CASES)
def test_forward_model_matches_reference(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    for i in range(z["M"].shape[1]):
        assert np.array_equal(mo.forward_model(z["G"], z["M"][:, i]), z["synth"][i])  # same op order: bit-exact


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("metric", METRICS)
@pytest.mark.parametrize("norm", [False, True])
@pytest.mark.parametrize("allat", [False, True])
def test_similarity_matches_reference(name, metric, norm, allat):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    ref = z["sim_%s_n%d_a%d" % (metric.replace("-", ""), norm, allat)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = np.array([mo.compare_synth_to_real_waveforms(z["d"], z["synth"][i], metric, norm, allat)
                        for i in range(len(ref))])
    # FFT-based correlate in the reference vs a direct sum here: ~1e-15 relative
    assert np.allclose(got, ref, rtol=1e-11, atol=1e-300), np.abs(got - ref).max()
    if metric == "gau" and not allat:
        assert not ref.any()  # the reference's per-trace 'gau' quirk: always 0


@pytest.mark.parametrize("name", CASES)
def test_driver_likelihood_and_posterior(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    MTs = z["drv_MTs"]
    sims, like = mo.score_samples(z["G"], z["d"], MTs, "VR", False, False)
    assert np.allclose(like, z["drv_MTp_absolute"], rtol=1e-12)
    assert np.allclose(mo.posterior(like), z["drv_MTp"], rtol=1e-12)
    assert abs(mo.posterior(like).sum() - 1.0) < 1e-12


def test_philox_known_answer_vectors():
    """Philox4x32-10 (the device sampler's generator) against Random123's published vectors."""
    for ctr, key, want in mo.PHILOX_KAT:
        got = mo.philox4x32_10(*[np.array([c]) for c in ctr], key[0], key[1])
        assert tuple(int(g[0]) for g in got) == want


def test_device_deviates_are_counter_based_and_well_distributed():
    a = mo.device_sampler_deviates("single_force_crack_no_coupling", 11, 0, 4096)
    b = mo.device_sampler_deviates("single_force_crack_no_coupling", 11, 1000, 100)
    for key in a:  # any index range reproduces the same per-sample deviates
        assert np.array_equal(a[key][1000:1100], b[key])
    z = mo.device_sampler_deviates("full_mt", 3, 0, 200000)["z6"]
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01 and abs((z ** 4).mean() - 3.0) < 0.1
    assert np.abs(np.corrcoef(z.T) - np.eye(6)).max() < 0.01
    c = mo.device_sampler_deviates("DC_crack_couple", 3, 0, 200000)
    for key, lo in (("u_theta", -1.0), ("r_phi", 0.0), ("r_quadrant", 0.0), ("frac", 0.0)):
        assert lo <= c[key].min() and c[key].max() < 1.0 and abs(c[key].mean() - (lo + 1.0) / 2) < 0.01
    assert not np.array_equal(mo.device_sampler_deviates("DC", 1, 0, 8)["z3"],
                              mo.device_sampler_deviates("DC", 2, 0, 8)["z3"])


def test_two_media_branch_vs_reference_golden(monkeypatch):
    """The worker's two-media branch, one fraction per phase type (:715-727, driver rows :855-861), against the
    reference's own output: (a) the oracle's restatement of the per-sample mixture, (b) the product's enlargement
    of it into ONE linear problem (`mixed_media_problem`), scored by the oracle here (no GPU in this suite), and
    (c) the product's driver with the reference's random stream."""
    import random
    from full_waveform_inversion_amd import samplers, source_inversion as si
    g = np.load(os.path.join(GOLDEN, "ref_multimedia.npz"))
    labels = [str(x) for x in g["labels"]]
    def oracle_score(d, G, M, metric, norm, allat, device=0):  # si.score_samples' contract, computed by the oracle
        sim, like = mo.score_samples(G, d, M, metric, norm, allat)
        return sim, like, mo.posterior(like)
    monkeypatch.setattr(si, "score_samples", oracle_score)
    for typ in ("full_mt", "DC_single_force_couple"):
        G2, d, MTs_ref = g["G2_" + typ], g["d_" + typ], g["MTs_" + typ]
        n = G2.shape[1]
        nextra = 4 if typ in samplers.COUPLED_TYPES else 3
        M, fr = MTs_ref[:n], MTs_ref[-3:].T          # samples (already scaled) and the (N, 3) fractions
        # (a) per-sample mixture, restated
        like_a = np.array([mo.likelihood(mo.compare_synth_to_real_waveforms(
            d, mo.forward_model(mo.mixed_media_greens(G2, dict(zip(si.PHASE_CLASSES, fr[i])), labels), M[:, i]),
            "VR", False, False)) for i in range(M.shape[1])])
        assert np.allclose(like_a, g["MTp_absolute_" + typ], rtol=1e-11, atol=0)
        # (b) the enlarged linear problem
        G_ext, expand = si.mixed_media_problem(G2, fr, labels)
        sim_b, like_b = mo.score_samples(G_ext, d, expand(M), "VR", False, False)
        assert np.allclose(like_b, g["MTp_absolute_" + typ], rtol=1e-11, atol=0)
        # (c) the driver, reference random stream
        seed = int(g["seed_" + typ])
        np.random.seed(seed)
        random.seed(seed)
        MTs, MTp, MTp_abs = si.perform_monte_carlo_sampled_waveform_inversion(
            d, G2, num_samples=M.shape[1], M_amplitude=0.9, inversion_type=typ, comparison_metric="VR",
            perform_normallised_waveform_inversion=False, compare_all_waveforms_simultaneously=False,
            reference_stream=True, invert_for_ratio_of_multiple_media_greens_func_switch=True,
            green_func_phase_labels=labels, num_phase_types_for_media_ratios=3)
        assert MTs.shape == MTs_ref.shape == (n + nextra, M.shape[1])
        assert np.allclose(MTs, MTs_ref, rtol=1e-12, atol=1e-14)
        assert np.allclose(MTp, g["MTp_" + typ], rtol=1e-10) and np.allclose(MTp_abs, g["MTp_absolute_" + typ], rtol=1e-10)
    # one fraction per sample: the reference cannot run this branch past its first sample (A-8); the enlargement is
    # checked against the evident mixture (1 - f) G_1 + f G_2
    rng = np.random.default_rng(3)
    f = rng.uniform(0, 1, 7)
    Mx = rng.standard_normal((n, 7))
    G_ext, expand = si.mixed_media_problem(G2, f)
    for i in range(7):
        Gm = (1 - f[i]) * G2[..., 0] + f[i] * G2[..., 1]
        assert np.allclose(mo.forward_model(G_ext, expand(Mx)[:, i]), mo.forward_model(Gm, Mx[:, i]), rtol=1e-12, atol=1e-14)
