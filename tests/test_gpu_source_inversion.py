"""GPU parity of the reference's REAL hot loop (SURVEY s.8f-2) -- PINNED to reference code.

Golden vectors in tests/golden/ref_mc_*.npz were produced by the reference's own functions
(tests/golden/make_reference_golden.py); the GPU path must reproduce them, and the pinned
oracle (oracle/mc_oracle.py) on fresh seeded inputs incl. the reference's shipped shape
(k = 21 traces, n = 9 components)."""
import os
import warnings

import numpy as np
import pytest

from full_waveform_inversion_amd import FwiError, source_inversion as si
from oracle import mc_oracle as mo

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["ref_mc_fullmt", "ref_mc_force", "ref_mc_dc"]
METRICS = ["VR", "CC", "PCC", "CC-shift", "gau"]
TOL = 1e-9  # fp64 both sides; moments vs direct sums differ by round-off only


@pytest.mark.parametrize("name", CASES)
def test_forward_model_vs_reference_golden(gpu, name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    got = si.forward_model_batch(z["G"], z["M"])
    assert np.allclose(got, z["synth"], rtol=1e-13, atol=1e-14 * np.abs(z["synth"]).max())
    one = si.forward_model(z["G"], z["M"][:, 3:4])
    assert np.allclose(one, z["synth"][3], rtol=1e-13, atol=1e-14 * np.abs(z["synth"]).max())


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("metric", METRICS)
@pytest.mark.parametrize("norm", [False, True])
@pytest.mark.parametrize("allat", [False, True])
def test_similarity_vs_reference_golden(gpu, name, metric, norm, allat):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    ref = z["sim_%s_n%d_a%d" % (metric.replace("-", ""), norm, allat)]
    sim, like, post = si.score_samples(z["d"], z["G"], z["M"], metric, norm, allat)
    assert np.allclose(sim, ref, rtol=TOL, atol=1e-12), np.abs(sim - ref).max()
    assert np.allclose(like, np.exp(-(1.0 - ref) / 2.0), rtol=TOL)
    assert abs(post.sum() - 1.0) < 1e-12


@pytest.mark.parametrize("name", CASES)
def test_driver_vs_reference_golden(gpu, name):
    """Posterior and likelihoods of the reference's own driver run (one process, seeded)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    MTs, MTp, MTp_abs = si.perform_monte_carlo_sampled_waveform_inversion(
        z["d"], z["G"], comparison_metric="VR", perform_normallised_waveform_inversion=False,
        compare_all_waveforms_simultaneously=False, MTs=z["drv_MTs"])
    assert np.allclose(MTp_abs, z["drv_MTp_absolute"], rtol=TOL)
    assert np.allclose(MTp, z["drv_MTp"], rtol=TOL)


@pytest.mark.parametrize("metric", METRICS)
def test_shipped_shape_vs_pinned_oracle(gpu, metric):
    """k = 21, n = 9, t = 512 (the reference's shipped configuration), 37 samples (ragged batch)."""
    rng = np.random.default_rng(3)
    k, n, t, N = 21, 9, 512, 37
    G = np.cumsum(rng.standard_normal((k, n, t)), axis=2) * np.hanning(t)
    Mt = rng.standard_normal(n)
    d = np.einsum("kjt,j->kt", G, Mt) + 0.1 * G.std() * rng.standard_normal((k, t))
    Ms = np.concatenate([rng.standard_normal((n, N - 1)), Mt[:, None]], axis=1)
    for norm in (False, True):
        for allat in (False, True):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                ref = np.array([mo.compare_synth_to_real_waveforms(d, mo.forward_model(G, Ms[:, i]), metric,
                                                                   norm, allat) for i in range(N)])
            sim = si.score_samples(d, G, Ms, metric, norm, allat)[0]
            assert np.allclose(sim, ref, rtol=TOL, atol=1e-12), (norm, allat, np.abs(sim - ref).max())


@pytest.mark.parametrize("n", [3, 6, 9, 5, 1, 8, 11])
@pytest.mark.parametrize("t", [2, 3, 4, 5, 6, 7, 9, 33])
def test_short_and_odd_traces_both_kernels(gpu, n, t):
    """Group / tail handling of the lane-per-sample kernel (n <= 9: two-row groups in ping-pong, so every residue
    of (t - 1) mod 4 is hit; 3 / 6 / 9 hand-written stages, the others the generic one) and the moment kernel
    (n = 11), ragged sample counts."""
    rng = np.random.default_rng(100 * n + t)
    k, N = 4, 300
    G = rng.standard_normal((k, n, t))
    Ms = rng.standard_normal((n, N))
    d = np.einsum("kjt,j->kt", G, Ms[:, 7]) + 0.3 * rng.standard_normal((k, t))
    for metric in ("VR", "CC", "CC-shift", "gau"):
        for norm, allat in ((False, False), (True, True)):
            if metric == "gau" and not allat:
                continue
            nref = 12 if metric == "CC-shift" else N  # the oracle's CC-shift manages ~50 samples/s
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                ref = np.array([mo.compare_synth_to_real_waveforms(d, mo.forward_model(G, Ms[:, i]), metric,
                                                                   norm, allat) for i in range(nref)])
            for cnt in (N, 1):
                sim = si.score_samples(d, G, Ms[:, :cnt], metric, norm, allat)[0]
                m = min(cnt, nref)
                assert sim.shape == (cnt,)
                assert np.allclose(sim[:m], ref[:m], rtol=1e-8, atol=1e-11, equal_nan=True), \
                    (metric, norm, allat, cnt, np.nanmax(np.abs(sim[:m] - ref[:m])))


@pytest.mark.parametrize("typ", si.samplers.INVERSION_TYPES)
def test_driver_every_inversion_type_vs_reference_golden(gpu, typ):
    """One-process runs of the reference's driver for each of its seven inversion types
    (tests/golden/ref_samplers.npz): same seeds -> same samples, amplitude-fraction row, posterior
    and likelihoods."""
    import random
    z = np.load(os.path.join(GOLDEN, "ref_samplers.npz"))
    seed = int(z["seed_" + typ]) + 7
    np.random.seed(seed)
    random.seed(seed)
    ref_MTs = z["drv_MTs_" + typ]
    MTs, MTp, MTp_abs = si.perform_monte_carlo_sampled_waveform_inversion(
        z["drv_d_" + typ], z["drv_G_" + typ], num_samples=ref_MTs.shape[1], M_amplitude=0.7, inversion_type=typ,
        comparison_metric="PCC", perform_normallised_waveform_inversion=True,
        compare_all_waveforms_simultaneously=False, reference_stream=True)
    assert MTs.shape == ref_MTs.shape
    assert np.allclose(MTs, ref_MTs, rtol=0, atol=1e-13, equal_nan=True)
    ok = np.isfinite(z["drv_MTp_absolute_" + typ])
    assert ok.all() or typ == "single_force_crack_no_coupling"  # the arccos sampler may emit NaN (Appendix A-2)
    assert np.allclose(MTp_abs[ok], z["drv_MTp_absolute_" + typ][ok], rtol=TOL, atol=1e-12)
    if ok.all():
        assert np.allclose(MTp, z["drv_MTp_" + typ], rtol=TOL, atol=1e-12)


@pytest.mark.parametrize("typ", si.samplers.INVERSION_TYPES)
def test_device_sampler_vs_oracle_deviates_and_reference_maps(gpu, typ):
    """mc_sample_kernel == the reference's sampler map (samplers.py, pinned to reference goldens) applied to
    the oracle's restatement of the device generator (Philox4x32-10 + Box-Muller, KAT-checked)."""
    seed, first, N, amp = 0x1234567890ABCDEF, (1 << 32) - 100, 777, 1.7   # index range crossing 2^32
    M, frac = si.sample_on_device(typ, N, seed, first, amp)
    with np.errstate(all="ignore"):
        ref_M, ref_frac = si.samplers.from_deviates(typ, mo.device_sampler_deviates(typ, seed, first, N))
    assert M.shape == ref_M.shape
    bad = ~np.isfinite(ref_M).all(axis=0)  # arccos quirk of the single-force-crack sampler (Appendix A-2)
    assert bad.mean() < 0.01 and np.array_equal(bad, ~np.isfinite(M).all(axis=0))
    assert np.allclose(M[:, ~bad], amp * ref_M[:, ~bad], rtol=0, atol=2e-11), np.abs(M - amp * ref_M)[:, ~bad].max()
    if ref_frac is not None:
        assert np.allclose(frac, ref_frac, rtol=0, atol=1e-15)
    # a sub-range regenerates the same samples
    M2, _ = si.sample_on_device(typ, 50, seed, first + 300, amp)
    assert np.array_equal(M2, M[:, 300:350], equal_nan=True)


@pytest.mark.parametrize("typ", ["full_mt", "DC_single_force_couple", "DC_crack_couple"])
def test_invert_on_device_equals_sample_then_score(gpu, typ):
    rng = np.random.default_rng(8)
    n, k, t, N = si.samplers.NUM_COMPONENTS[typ], 5, 96, 1000
    G = rng.standard_normal((k, n, t))
    d = np.einsum("kjt,j->kt", G, rng.standard_normal(n))
    M, frac, sim, like, post = si.invert_on_device(d, G, N, typ, seed=5, M_amplitude=0.5, comparison_metric="CC",
                                                   perform_normallised_waveform_inversion=True,
                                                   compare_all_waveforms_simultaneously=False)
    M2, frac2 = si.sample_on_device(typ, N, 5, 0, 0.5)
    assert np.array_equal(M, M2) and np.array_equal(frac, frac2)
    sim2, like2, post2 = si.score_samples(d, G, M, "CC", True, False)
    assert np.array_equal(sim, sim2) and np.array_equal(like, like2) and np.array_equal(post, post2)
    none = si.invert_on_device(d, G, N, typ, seed=5, M_amplitude=0.5, comparison_metric="CC",
                               perform_normallised_waveform_inversion=True,
                               compare_all_waveforms_simultaneously=False, return_samples=False)
    assert none[0] is None and none[1] is None and np.array_equal(none[2], sim)
    MTs, MTp, MTp_abs = si.perform_monte_carlo_sampled_waveform_inversion(
        d, G, N, 0.5, typ, "CC", True, False, seed=5)
    assert MTs.shape == (n + (typ in si.samplers.COUPLED_TYPES), N) and np.array_equal(MTs[:n], M)
    assert np.array_equal(MTp, post) and np.array_equal(MTp_abs, like)
    with pytest.raises(FwiError):  # component count must match the type
        si.invert_on_device(d, G[:, :2], N, typ)


def test_two_ranks_share_one_run(gpu, tmp_path):
    """Two processes on this GPU (sample-index blocks + one scalar all-reduce over the stdlib control plane) reproduce the
    single-process device run bit for bit."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    import _mc_dist_worker as w
    from test_host_logic import _run_mc_world2
    res = _run_mc_world2(tmp_path, "gpu")
    d, G, N, typ = w.problem()
    M, frac, sim, like, post = si.invert_on_device(d, G, N, typ, 9, 0, 1.5, "VR", False, False)
    assert np.array_equal(np.hstack([r["M"] for r in res]), np.vstack((M, frac)))
    assert np.array_equal(np.concatenate([r["like"] for r in res]), like)
    assert np.allclose(np.concatenate([r["post"] for r in res]), post, rtol=1e-12, atol=0)


def test_plan_blocks_equal_one_call(gpu):
    """fwi_mc_plan_*: a run cut into blocks (device state kept between them) gives the samples, scores and
    -- after the caller's renormalisation with the blocks' likelihood sums -- the posterior of one call."""
    rng = np.random.default_rng(4)
    typ, k, t, N = "DC_single_force_no_coupling", 6, 70, 2500
    n = si.samplers.NUM_COMPONENTS[typ]
    G = rng.standard_normal((k, n, t))
    d = np.einsum("kjt,j->kt", G, rng.standard_normal(n)) + 0.1 * rng.standard_normal((k, t))
    M, frac, sim, like, post = si.invert_on_device(d, G, N, typ, 3, 100, 0.8, "PCC", True, False)
    got = []
    with si.MonteCarloPlan(d, G, max_samples=1024) as plan:
        for first in range(0, N, 1024):
            got.append(plan.invert(typ, min(1024, N - first), 3, 100 + first, 0.8, "PCC", True, False))
        assert np.array_equal(np.hstack([g[0] for g in got]), M) and np.array_equal(np.concatenate([g[1] for g in got]), frac)
        assert np.array_equal(np.concatenate([g[2] for g in got]), sim)
        like_all, total = np.concatenate([g[3] for g in got]), sum(g[4] for g in got)
        assert np.array_equal(like_all, like) and np.allclose(like_all / total, post, rtol=1e-12, atol=0)
        # given samples through the same plan, another metric / mode without re-creating it
        s2, l2, tot2 = plan.score(M[:, :900], "VR", False, True)
        ref = si.score_samples(d, G, M[:, :900], "VR", False, True)
        assert np.array_equal(s2, ref[0]) and np.array_equal(l2, ref[1]) and abs(tot2 - ref[1].sum()) < 1e-9 * tot2
        with pytest.raises(FwiError):
            plan.invert(typ, 1025)                       # beyond the plan's capacity
        with pytest.raises(FwiError):
            plan.invert("full_mt", 10)                   # 6 components against 9-component Green's functions


def test_best_of_a_run_too_large_to_return(gpu):
    rng = np.random.default_rng(6)
    typ, k, t, N = "full_mt", 5, 64, 300000
    G = rng.standard_normal((k, 6, t))
    Mt = si.sample_on_device(typ, 1, 17, 123456)[0]      # sample 123456 of the stream is the truth
    d = np.einsum("kjt,j->kt", G, Mt[:, 0])
    idx, MTs, MTp, total = si.monte_carlo_best_of(d, G, N, typ, seed=17, comparison_metric="VR",
                                                  perform_normallised_waveform_inversion=False,
                                                  compare_all_waveforms_simultaneously=False, block=65536, keep=50)
    assert idx[0] == 123456 and np.allclose(MTs[:, 0], Mt[:, 0]) and MTs.shape == (6, 50)
    assert np.all(np.diff(MTp) <= 0) and 0 < MTp.sum() < 1
    _, _, _, like, post = si.invert_on_device(d, G, N, typ, 17, 0, 1.0, "VR", False, False, return_samples=False)
    assert abs(total - like.sum()) < 1e-9 * total and np.allclose(MTp, np.sort(post)[::-1][:50], rtol=1e-10)


def test_million_samples_best_is_truth(gpu):
    """Size-independent property at production scale: 2^20 samples, the planted source scores highest."""
    rng = np.random.default_rng(0)
    k, n, t, N = 21, 6, 256, 1 << 20
    G = rng.standard_normal((k, n, t))
    Mt = si.random_full_mt(1, rng)
    d = np.einsum("kjt,j->kt", G, Mt[:, 0])
    Ms = si.random_full_mt(N, rng)
    Ms[:, 12345] = Mt[:, 0]
    sim, like, post, ms = si.score_samples(d, G, Ms, "VR", False, False, return_timing=True)
    assert np.argmax(sim) == 12345 and abs(sim[12345] - 1.0) < 1e-12
    assert abs(post.sum() - 1.0) < 1e-9 and np.argmax(post) == 12345
    assert N / (ms * 1e-3) > 1e6  # the reference manages ~1e3 samples/s per core


def test_unnormalised_probability_of_a_given_solution(gpu):
    z = np.load(os.path.join(GOLDEN, "ref_mc_fullmt.npz"))
    for i in (0, 7, 23):
        p = si.get_unnormallised_prob_for_specific_soln(z["d"], z["G"], z["M"][:, i:i + 1], "VR", True, False)
        assert abs(p - z["sim_VR_n1_a0"][i]) <= TOL


def test_errors(gpu):
    G, d = np.ones((2, 3, 8)), np.ones((2, 8))
    with pytest.raises(ValueError):
        si.score_samples(d, G, np.ones((4, 5)))
    with pytest.raises(ValueError):
        si.score_samples(d, G, np.ones((3, 5)), "nope")
    with pytest.raises(FwiError):
        si.score_samples(d, G, np.ones((3, 5)), device=99)
    with pytest.raises(ValueError):
        si.sample_on_device("nope", 4)
    with pytest.raises(FwiError):
        si.sample_on_device("DC", 0)
    with pytest.raises(FwiError):
        si.sample_on_device("DC", 4, first_sample=-1)
    with pytest.raises(FwiError, match="too large"):   # the moment kernel (n > 9) keeps k * n moments in LDS ...
        si.score_samples(np.ones((400, 8)), np.ones((400, 12, 8)), np.ones((12, 3)))


def test_many_traces_take_the_lane_kernel_without_an_lds_limit(gpu):
    """... the lane-per-sample kernel (n <= 9) has no such limit."""
    rng = np.random.default_rng(2)
    k, n, t, N = 400, 3, 16, 70
    G = rng.standard_normal((k, n, t))
    Ms = rng.standard_normal((n, N))
    d = np.einsum("kjt,j->kt", G, Ms[:, 1]) + 0.2 * rng.standard_normal((k, t))
    sim = si.score_samples(d, G, Ms, "PCC", True, False)[0]
    ref = mo.score_samples(G, d, Ms, "PCC", True, False)[0]
    assert np.allclose(sim, ref, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("typ", ["full_mt", "DC_single_force_couple"])
def test_two_media_branch_on_gpu_vs_reference_golden(gpu, typ):
    """invert_for_ratio_of_multiple_media_greens_func_switch (:715-727, :855-861) through the GPU scoring kernel
    (the enlarged 6n-component problem takes the n > 9 moment kernel): samples, fraction rows, posterior and
    likelihoods of the reference's own run, reproduced with its random stream."""
    import random
    from full_waveform_inversion_amd import source_inversion as si
    g = np.load(os.path.join(GOLDEN, "ref_multimedia.npz"))
    labels = [str(x) for x in g["labels"]]
    seed = int(g["seed_" + typ])
    np.random.seed(seed)
    random.seed(seed)
    MTs, MTp, MTp_abs = si.perform_monte_carlo_sampled_waveform_inversion(
        g["d_" + typ], g["G2_" + typ], num_samples=g["MTs_" + typ].shape[1], M_amplitude=0.9, inversion_type=typ,
        comparison_metric="VR", perform_normallised_waveform_inversion=False,
        compare_all_waveforms_simultaneously=False, reference_stream=True,
        invert_for_ratio_of_multiple_media_greens_func_switch=True, green_func_phase_labels=labels,
        num_phase_types_for_media_ratios=3)
    assert np.allclose(MTs, g["MTs_" + typ], rtol=1e-12, atol=1e-14)
    assert np.allclose(MTp_abs, g["MTp_absolute_" + typ], rtol=1e-9, atol=0)
    assert np.allclose(MTp, g["MTp_" + typ], rtol=1e-9, atol=0) and abs(MTp.sum() - 1.0) < 1e-12
    # the default (seeded, non-reference) stream and the one-fraction variant run and normalise
    MTs2, MTp2, _ = si.perform_monte_carlo_sampled_waveform_inversion(
        g["d_" + typ], g["G2_" + typ], num_samples=500, inversion_type=typ, comparison_metric="PCC", seed=4,
        invert_for_ratio_of_multiple_media_greens_func_switch=True)
    assert MTs2.shape[0] == g["MTs_" + typ].shape[0] - 2 and abs(MTp2.sum() - 1.0) < 1e-12
    assert (MTs2[-1] >= 0).all() and (MTs2[-1] <= 1).all()
