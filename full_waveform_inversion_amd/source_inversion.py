"""GPU drop-in for the reference's REAL hot loop.

Mirrors the reference's function names and argument meaning for the source-inversion path:
``forward_model(green_func_array, M)`` (full_waveform_inversion.py:253),
``compare_synth_to_real_waveforms(...)`` (:584) and the scoring part of
``perform_monte_carlo_sampled_waveform_inversion`` (:786).  The per-sample loop body
(forward model + similarity + likelihood, :713-774) runs as one fused HIP kernel over all
samples; the seven random source samplers (:282-510) live in ``samplers.py`` (batched on the
host, reproducing the reference's stream on request).  fp64 like the reference.
There is no CPU fallback: without the HIP library / a GPU these raise FwiError.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib, samplers


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def forward_model_batch(green_func_array, MTs, device=0):
    """Synthetics for every column of ``MTs (n, N)``: array ``(N, k, t)``."""
    G = _f64(green_func_array)
    M = _f64(MTs)
    if G.ndim != 3 or M.ndim != 2 or M.shape[0] != G.shape[1]:
        raise ValueError("green_func_array must be (k, n, t) and MTs (n, N)")
    k, n, t = G.shape
    out = np.empty((M.shape[1], k, t))
    _lib.check(None, _lib.load().fwi_mc_forward(device, k, n, t, M.shape[1], _p(G), _p(M), _p(out)))
    return out


def forward_model(green_func_array, M, device=0):
    """``synth[k, t] = sum_j G[k, j, t] * M[j]`` -- same contract as the reference's function."""
    M = _f64(M).reshape(-1, 1)
    return forward_model_batch(green_func_array, M, device)[0]


def score_samples(real_data_array, green_func_array, MTs, comparison_metric="VR",
                  perform_normallised_waveform_inversion=True, compare_all_waveforms_simultaneously=True,
                  device=0, return_timing=False):
    """Similarity of every sample's synthetic to the data: the reference's steps 4-5 for given samples.

    Returns ``(similarity (N,), likelihood (N,), posterior (N,))`` [+ kernel milliseconds].
    """
    G, d, M = _f64(green_func_array), _f64(real_data_array), _f64(MTs)
    if G.ndim != 3 or d.shape != (G.shape[0], G.shape[2]) or M.ndim != 2 or M.shape[0] != G.shape[1]:
        raise ValueError("shapes must be G (k, n, t), data (k, t), MTs (n, N)")
    if comparison_metric not in _lib.MC_METRICS:
        raise ValueError("comparison_metric must be one of %s" % sorted(_lib.MC_METRICS))
    k, n, t = G.shape
    N = M.shape[1]
    sim, like, post = np.empty(N), np.empty(N), np.empty(N)
    ms = C.c_double(0.0)
    _lib.check(None, _lib.load().fwi_mc_score(
        device, k, n, t, N, _p(G), _p(d), _p(M), _lib.MC_METRICS[comparison_metric],
        int(bool(perform_normallised_waveform_inversion)), int(bool(compare_all_waveforms_simultaneously)),
        _p(sim), _p(like), _p(post), C.byref(ms)))
    return (sim, like, post, ms.value) if return_timing else (sim, like, post)


def compare_synth_to_real_waveforms(real_data_array, green_func_array, M, comparison_metric,
                                    perform_normallised_waveform_inversion=True,
                                    compare_all_waveforms_simultaneously=True, device=0):
    """Similarity for ONE source vector.  Differs from the reference's signature in taking the
    Green's functions and the source instead of a precomputed synthetic: the synthetic never
    leaves the GPU."""
    return float(score_samples(real_data_array, green_func_array, _f64(M).reshape(-1, 1), comparison_metric,
                               perform_normallised_waveform_inversion, compare_all_waveforms_simultaneously,
                               device)[0][0])


def get_unnormallised_prob_for_specific_soln(real_data_array, green_func_array, MT_specific_soln, comparison_metric,
                                            perform_normallised_waveform_inversion=True,
                                            compare_all_waveforms_simultaneously=True, device=0):
    """Unnormalised probability (= similarity) of one given solution: the core of the reference's
    third script (unnormallised_probability_retrieval_from_full_waveform_soln.py:222-232), i.e.
    forward model + comparison, both on the GPU."""
    return compare_synth_to_real_waveforms(real_data_array, green_func_array, MT_specific_soln, comparison_metric,
                                           perform_normallised_waveform_inversion,
                                           compare_all_waveforms_simultaneously, device)


def sample_on_device(inversion_type, num_samples, seed=0, first_sample=0, M_amplitude=1.0, device=0):
    """``num_samples`` sources of ``inversion_type`` drawn by the device sampler (``fwi_mc_sample``):
    the reference's generate_random_* maps (:282-510) applied to counter-based deviates, sample
    ``i`` a pure function of ``(seed, first_sample + i)``.  Returns ``(MTs (n, N), amp_frac (N,))``."""
    if inversion_type not in _lib.MC_INVERSION_TYPES:
        raise ValueError("inversion_type must be one of %s" % sorted(_lib.MC_INVERSION_TYPES))
    n = samplers.NUM_COMPONENTS[inversion_type]
    M, frac = np.empty((n, int(num_samples))), np.empty(int(num_samples))
    _lib.check(None, _lib.load().fwi_mc_sample(device, _lib.MC_INVERSION_TYPES[inversion_type], int(seed),
                                               int(first_sample), int(num_samples), float(M_amplitude),
                                               _p(M), _p(frac)))
    return M, frac


def invert_on_device(real_data_array, green_func_array, num_samples, inversion_type="full_mt", seed=0,
                     first_sample=0, M_amplitude=1.0, comparison_metric="VR",
                     perform_normallised_waveform_inversion=True, compare_all_waveforms_simultaneously=True,
                     return_samples=True, device=0, return_timing=False):
    """Steps 3-7 of the reference's worker (:713-774) entirely on the GPU (``fwi_mc_invert``): draw,
    forward-model, compare, likelihood.  Returns ``(MTs or None, amp_frac or None, similarity,
    likelihood, posterior)`` [+ kernel ms]; with ``return_samples=False`` no sample crosses PCIe
    (any of them can be regenerated later from ``(seed, index)`` with ``sample_on_device``)."""
    G, d = _f64(green_func_array), _f64(real_data_array)
    if G.ndim != 3 or d.shape != (G.shape[0], G.shape[2]):
        raise ValueError("shapes must be G (k, n, t), data (k, t)")
    if inversion_type not in _lib.MC_INVERSION_TYPES:
        raise ValueError("inversion_type must be one of %s" % sorted(_lib.MC_INVERSION_TYPES))
    if comparison_metric not in _lib.MC_METRICS:
        raise ValueError("comparison_metric must be one of %s" % sorted(_lib.MC_METRICS))
    k, n, t = G.shape
    N = int(num_samples)
    M = np.empty((n, N)) if return_samples else None
    frac = np.empty(N) if return_samples else None
    sim, like, post = np.empty(N), np.empty(N), np.empty(N)
    ms = C.c_double(0.0)
    _lib.check(None, _lib.load().fwi_mc_invert(
        device, _lib.MC_INVERSION_TYPES[inversion_type], int(seed), int(first_sample), N, float(M_amplitude),
        k, n, t, _p(G), _p(d), _lib.MC_METRICS[comparison_metric],
        int(bool(perform_normallised_waveform_inversion)), int(bool(compare_all_waveforms_simultaneously)),
        _p(M) if return_samples else None, _p(frac) if return_samples else None, _p(sim), _p(like), _p(post),
        C.byref(ms)))
    out = (M, frac, sim, like, post)
    return out + (ms.value,) if return_timing else out


def random_full_mt(num_samples, rng):
    """Unit 6-vectors uniform on the 5-sphere (generate_random_MT, :282-293)."""
    return samplers.draw("full_mt", num_samples, rng)[0]


def random_single_force(num_samples, rng):
    """Unit 3-vectors uniform on the sphere (generate_random_single_force_vector, :320-331)."""
    return samplers.draw("single_force", num_samples, rng)[0]


def perform_monte_carlo_sampled_waveform_inversion(real_data_array, green_func_array, num_samples=1000,
                                                   M_amplitude=1.0, inversion_type="full_mt",
                                                   comparison_metric="CC",
                                                   perform_normallised_waveform_inversion=True,
                                                   compare_all_waveforms_simultaneously=True, MTs=None,
                                                   seed=0, device=0, reference_stream=False,
                                                   return_absolute_similarity_values_switch=True,
                                                   sampler="device",
                                                   invert_for_ratio_of_multiple_media_greens_func_switch=False,
                                                   green_func_phase_labels=(), num_phase_types_for_media_ratios=0):
    """The reference's driver (:786-870) with the sample loop on the GPU.

    With ``invert_for_ratio_of_multiple_media_greens_func_switch`` the call is
    :func:`perform_monte_carlo_sampled_waveform_inversion_multiple_media` (Green's functions ``(k, n, t, 2)``).

    ``inversion_type`` is any of the reference's seven (:740-760): ``full_mt``, ``DC``,
    ``single_force``, ``DC_single_force_couple``, ``DC_single_force_no_coupling``,
    ``DC_crack_couple``, ``single_force_crack_no_coupling``.  Samples are drawn on the GPU (``sampler="device"``:
    ``fwi_mc_invert``, counter-based deviates keyed by ``seed``), on the host by ``samplers.draw``
    from ``default_rng(seed)`` (``sampler="host"``), or -- ``reference_stream=True`` -- on the host
    from the global ``numpy.random`` / ``random`` generators in the reference's order, which
    reproduces a one-process run of the reference seeded the same way.  ``MTs (n, N)`` may be
    supplied instead.

    Returns ``(MTs, MTp, MTp_absolute)`` like the reference: the samples (scaled by ``M_amplitude``,
    with the sampler's amplitude fraction appended as an extra row for the four coupled types,
    :852-853), the posterior ``L / sum L`` (:847-848) and the likelihoods ``exp(-(1-s)/2)`` (:774;
    ``[]`` when ``return_absolute_similarity_values_switch`` is false, :866-869).
    """
    if sampler not in ("device", "host"):
        raise ValueError("sampler must be 'device' or 'host'")
    if invert_for_ratio_of_multiple_media_greens_func_switch:
        return perform_monte_carlo_sampled_waveform_inversion_multiple_media(
            real_data_array, green_func_array, num_samples, M_amplitude, inversion_type, comparison_metric,
            perform_normallised_waveform_inversion, compare_all_waveforms_simultaneously, green_func_phase_labels,
            num_phase_types_for_media_ratios, seed, device, reference_stream, return_absolute_similarity_values_switch)
    frac = None
    if MTs is None and sampler == "device" and not reference_stream:
        MTs, frac, _, like, post = invert_on_device(
            real_data_array, green_func_array, num_samples, inversion_type, seed, 0, M_amplitude, comparison_metric,
            perform_normallised_waveform_inversion, compare_all_waveforms_simultaneously, True, device)
        if inversion_type in samplers.COUPLED_TYPES:
            MTs = np.vstack((MTs, frac))
        return MTs, post, (like if return_absolute_similarity_values_switch else [])
    if MTs is None:
        rng = None if reference_stream else np.random.default_rng(seed)
        MTs, frac = samplers.draw(inversion_type, num_samples, rng, reference_stream)
        MTs = MTs * M_amplitude
    MTs = _f64(MTs)
    _, like, post = score_samples(real_data_array, green_func_array, MTs, comparison_metric,
                                  perform_normallised_waveform_inversion,
                                  compare_all_waveforms_simultaneously, device)
    if frac is not None:
        MTs = np.vstack((MTs, frac))
    return MTs, post, (like if return_absolute_similarity_values_switch else [])


PHASE_CLASSES = ("P", "S", "surface")  # the reference's three phase types (:718-720)


def mixed_media_problem(green_func_array_both_media, frac_medium_2, green_func_phase_labels=None):
    """The two-media mixture of the worker (:715-731) as ONE linear problem the scoring kernel already solves.

    Per sample the reference scores ``G = (1 - f) G_medium1 + f G_medium2`` -- with one fraction ``f`` for all traces,
    or (``num_phase_types_for_media_ratios > 0``) one per phase type, every trace taking the fraction of its label.
    The mixture is linear in the source, so with the Green's functions laid side by side, ``[G_1 | G_2]`` per phase
    class, the sample ``M`` with fractions ``f_p`` becomes the source ``[(1 - f_p) M ; f_p M]_p`` of an enlarged
    problem with ``2 n`` (one fraction) or ``6 n`` (three phase classes) components.  Returns
    ``(G_ext (k, n_ext, t), expand)`` with ``expand(MTs (n, N)) -> (n_ext, N)``.
    """
    G2 = _f64(green_func_array_both_media)
    if G2.ndim != 4 or G2.shape[3] != 2:
        raise ValueError("two-media Green's functions must be (k, n, t, 2)")
    k, n, t, _ = G2.shape
    f = _f64(frac_medium_2)
    if f.ndim == 1:  # one fraction per sample
        G_ext = np.concatenate((G2[..., 0], G2[..., 1]), axis=1)
        return G_ext, lambda M: np.vstack(((1.0 - f)[None, :] * M, f[None, :] * M))
    labels = list(green_func_phase_labels or [])
    if len(labels) != k or any(lab not in PHASE_CLASSES for lab in labels):
        raise ValueError("green_func_phase_labels must hold one of %s per trace" % (PHASE_CLASSES,))
    G_ext = np.zeros((k, 6 * n, t))
    for j, lab in enumerate(labels):
        c = PHASE_CLASSES.index(lab)
        G_ext[j, 2 * n * c:2 * n * c + n] = G2[j, :, :, 0]
        G_ext[j, 2 * n * c + n:2 * n * (c + 1)] = G2[j, :, :, 1]

    def expand(M):
        return np.vstack([blk for c in range(3) for blk in ((1.0 - f[:, c])[None, :] * M, f[:, c][None, :] * M)])
    return G_ext, expand


def perform_monte_carlo_sampled_waveform_inversion_multiple_media(
        real_data_array, green_func_array, num_samples=1000, M_amplitude=1.0, inversion_type="full_mt",
        comparison_metric="CC", perform_normallised_waveform_inversion=True,
        compare_all_waveforms_simultaneously=True, green_func_phase_labels=(), num_phase_types_for_media_ratios=0,
        seed=0, device=0, reference_stream=False, return_absolute_similarity_values_switch=True):
    """The driver with ``invert_for_ratio_of_multiple_media_greens_func_switch`` on (:786-870, worker :715-731):
    ``green_func_array`` is ``(k, n, t, 2)``, every sample also draws the fraction of medium 2 -- one per sample,
    or one per phase type (P, S, surface) -- and the fractions are appended to ``MTs`` as extra rows (:855-864).

    The sample loop is the same GPU scoring call on the enlarged linear problem of :func:`mixed_media_problem`.
    The reference's one-fraction branch overwrites its own Green's functions after the first sample (:731, SURVEY
    Appendix A-8) and cannot run past it; what is implemented is its evident intent, a fresh mixture per sample.
    ``reference_stream=True`` draws fractions and samples from the global ``numpy.random`` / ``random`` generators in
    the reference's per-sample order (fractions first, :718-720 / :730).
    """
    n_frac = 3 if num_phase_types_for_media_ratios > 0 else 1
    N = int(num_samples)
    frac = None
    if reference_stream:
        fr, cols, fracs = np.empty((N, n_frac)), [], []
        for i in range(N):
            fr[i] = [np.random.uniform(0.0, 1.0) for _ in range(n_frac)]
            M1, f1 = samplers.draw(inversion_type, 1, None, True)
            cols.append(M1)
            fracs.append(f1)
        MTs = np.hstack(cols) * M_amplitude
        frac = np.concatenate(fracs) if fracs[0] is not None else None
    else:
        rng = np.random.default_rng(seed)
        fr = rng.uniform(0.0, 1.0, (N, n_frac))
        MTs, frac = samplers.draw(inversion_type, N, rng, False)
        MTs = MTs * M_amplitude
    G_ext, expand = mixed_media_problem(green_func_array, fr if n_frac == 3 else fr[:, 0], green_func_phase_labels)
    _, like, post = score_samples(real_data_array, G_ext, expand(_f64(MTs)), comparison_metric,
                                  perform_normallised_waveform_inversion, compare_all_waveforms_simultaneously, device)
    if inversion_type in samplers.COUPLED_TYPES:
        MTs = np.vstack((MTs, frac))
    MTs = np.vstack([MTs] + [fr[:, c] for c in range(n_frac)])  # :855-864
    return MTs, post, (like if return_absolute_similarity_values_switch else [])


def partition_samples(num_samples, rank, world):
    """``(first, count)`` of ``rank``'s contiguous block of sample indices.  The reference gives
    every worker ``int(N / P)`` samples and leaves the last ``N mod P`` slots empty (:822, SURVEY
    Appendix A-7); here the remainder is spread over the first ranks so all ``N`` are drawn."""
    base, extra = divmod(int(num_samples), int(world))
    return rank * base + min(rank, extra), base + (1 if rank < extra else 0)


def perform_monte_carlo_sampled_waveform_inversion_sharded(real_data_array, green_func_array, num_samples,
                                                           rank, world, sum_over_ranks, M_amplitude=1.0,
                                                           inversion_type="full_mt", comparison_metric="CC",
                                                           perform_normallised_waveform_inversion=True,
                                                           compare_all_waveforms_simultaneously=True, seed=0,
                                                           device=0, return_samples=True):
    """The reference's multi-process driver (:816-848) as one process per GPU: rank ``r`` draws and
    scores the samples of its index block (the device sampler is counter-based, so the ``world``
    blocks together are exactly the samples a single GPU would draw from ``seed``), and the only
    exchange is the posterior's normaliser ``sum L`` -- ``sum_over_ranks(float) -> float``, one
    scalar all-reduce (``Rendezvous.allreduce`` or ``Engine.allreduce_f64``).

    Returns ``(first_index, MTs_local, MTp_local, MTp_absolute_local)``; ``MTp`` is normalised over
    ALL ranks' samples (:847-848 with ``p_model = 1 / N``).
    """
    first, count = partition_samples(num_samples, rank, world)
    M, frac, _, like, _ = invert_on_device(
        real_data_array, green_func_array, count, inversion_type, seed, first, M_amplitude, comparison_metric,
        perform_normallised_waveform_inversion, compare_all_waveforms_simultaneously, return_samples, device)
    total = float(sum_over_ranks(float(np.sum(like))))
    if return_samples and inversion_type in samplers.COUPLED_TYPES:
        M = np.vstack((M, frac))
    return first, M, like / total, like


def perform_inversion(real_data_array, green_func_array):
    """Least-squares source estimate (:242-251): stack the traces in time, solve ``G M = D``.
    Returns ``M (n, 1)``.  Host-side (one small ``lstsq``); the reference uses it for the amplitude
    ``|M|`` the Monte Carlo samples are scaled by (:1171-1172)."""
    d, G = _f64(real_data_array), _f64(green_func_array)
    D = d.reshape(-1, 1)                                  # [d_0; d_1; ...]
    A = np.transpose(G, (0, 2, 1)).reshape(-1, G.shape[1])  # rows (trace, time), columns = components
    return np.linalg.lstsq(A, D, rcond=-1)[0]             # rcond=-1: the legacy default the reference ran with


def get_synth_forward_model_most_likely_result(MTs, MTp, green_func_array, inversion_type, device=0,
                                               invert_for_ratio_of_multiple_media_greens_func_switch=False,
                                               green_func_phase_labels=(), num_phase_types_for_media_ratios=0):
    """Synthetic of the highest-posterior sample (:974-1020): the coupled types carry the amplitude
    fraction as an extra row, which is not a source component.  Two media (``green_func_array (k, n, t, 2)``): the
    sample's last row(s) are the fraction(s) of medium 2 -- one, or one per phase type (P, S, surface), every trace
    mixing its two Green's functions with the fraction of its label -- and the synthetic is formed on that mixture."""
    MTs = np.asarray(MTs)
    best = int(np.where(MTp == np.max(MTp))[0][0])
    coupled = inversion_type in samplers.COUPLED_TYPES
    if not invert_for_ratio_of_multiple_media_greens_func_switch:
        rows = slice(None, -1) if coupled else slice(None)
        return forward_model(green_func_array, MTs[rows, best], device)
    G2 = _f64(green_func_array)
    if num_phase_types_for_media_ratios > 0:
        frac = dict(zip(PHASE_CLASSES, MTs[-3:, best]))
        G = np.zeros(G2.shape[:3])
        for j, lab in enumerate(green_func_phase_labels):  # (traces beyond the labels stay zero, as in :989-992)
            G[j] = (1.0 - frac[lab]) * G2[j, :, :, 0] + frac[lab] * G2[j, :, :, 1]
        nextra = 4 if coupled else 3
    else:
        f = MTs[-1, best]
        G = (1.0 - f) * G2[..., 0] + f * G2[..., 1]
        nextra = 2 if coupled else 1
    return forward_model(G, MTs[:-nextra, best], device)


def run_multi_medium_inversion(datadir, outdir, real_data_fnames, MT_green_func_fnames,
                               single_force_green_func_fnames, data_labels, inversion_type,
                               perform_normallised_waveform_inversion, compare_all_waveforms_simultaneously,
                               num_samples, comparison_metric, manual_indices_time_shift_MT=(),
                               manual_indices_time_shift_SF=(), uid="event", stations=(), cut_phase_start_vals=(),
                               cut_phase_length=0, set_pre_time_shift_values_to_zero_switch=True,
                               only_save_non_zero_solns_switch=False, return_absolute_similarity_values_switch=False,
                               green_func_fnames_split_index=0, green_func_phase_labels=(), seed=0, device=0,
                               reference_stream=False, nlloc_hyp_filename=None):
    """The reference's two-media driver (:1037-1158): load both media's Green's functions, least-squares estimate on
    their 50 / 50 mixture (saved under ``<outdir>/least_squares_result``), Monte Carlo inversion for the source AND the
    fraction(s) of medium 2 on the GPU, result and best-fit waveforms pickled in the reference's layout.

    As written the reference cannot get past its least-squares block: it forms that block's "most likely" synthetic on
    the 4-D two-media array (:1084-1086), which ``forward_model`` cannot take, and its shipped ``__main__`` never
    passes the switch on.  Here that synthetic is formed on the same 50 / 50 mixture the
    estimate was made on; everything else follows the listed lines.  ``uid`` / ``stations`` replace the NonLinLoc file.
    Returns ``(MTs, MTp, MTp_absolute)`` with the fraction rows appended to ``MTs`` (:855-864).
    """
    from . import io
    if nlloc_hyp_filename:
        uid, stations = io.get_event_uid_and_station_data_MTFIT_FORMAT_from_nonlinloc_hyp_file(nlloc_hyp_filename)
    real, G2 = io.get_overall_real_and_green_func_data(
        datadir, real_data_fnames, MT_green_func_fnames, single_force_green_func_fnames, inversion_type,
        manual_indices_time_shift_MT, manual_indices_time_shift_SF, cut_phase_start_vals, cut_phase_length,
        set_pre_time_shift_values_to_zero_switch, invert_for_ratio_of_multiple_media_greens_func_switch=True,
        green_func_fnames_split_index=green_func_fnames_split_index)
    labels = list(green_func_phase_labels)
    if labels and len(labels) != G2.shape[0]:
        raise ValueError("green_func_phase_labels must hold one label per trace (%d traces, %d labels)"
                         % (G2.shape[0], len(labels)))  # (:1044-1047 print + exit)
    nphase = sum(1 for c in PHASE_CLASSES if labels.count(c) > 0)  # (:1050-1056)
    G_lsq = 0.5 * G2[..., 0] + 0.5 * G2[..., 1]  # frac_medium_2 = 0.5 (:1059-1060)
    M = perform_inversion(real, G_lsq)
    M_amplitude = float(np.sum(M ** 2) ** 0.5)
    lsq_sim = compare_synth_to_real_waveforms(real, G_lsq, M, comparison_metric, perform_normallised_waveform_inversion,
                                              compare_all_waveforms_simultaneously, device)
    lsq_dir = os.path.join(outdir, "least_squares_result")
    io.save_to_MTFIT_style_file(M, np.array([lsq_sim]), uid, inversion_type, lsq_dir, stations)
    io.save_specific_waveforms_to_file(real, forward_model(G_lsq, M, device), data_labels, uid, inversion_type, lsq_dir)
    MTs, MTp, MTp_absolute = perform_monte_carlo_sampled_waveform_inversion_multiple_media(
        real, G2, num_samples, M_amplitude, inversion_type, comparison_metric, perform_normallised_waveform_inversion,
        compare_all_waveforms_simultaneously, labels, nphase, seed=seed, device=device,
        reference_stream=reference_stream,
        return_absolute_similarity_values_switch=return_absolute_similarity_values_switch)
    if np.isnan(MTp[0]):
        raise FloatingPointError("sum of probabilities is zero: no adequate solution found (:1094-1096)")
    if only_save_non_zero_solns_switch:
        MTp, MTs = io.remove_zero_prob_results(MTp, MTs)
    io.save_to_MTFIT_style_file(MTs, MTp, uid, inversion_type, outdir, stations, MTp_absolute)
    best = get_synth_forward_model_most_likely_result(MTs, MTp, G2, inversion_type, device, True, labels, nphase)
    io.save_specific_waveforms_to_file(real, best, data_labels, uid, inversion_type, outdir)
    return MTs, MTp, MTp_absolute


def run(datadir, outdir, real_data_fnames, MT_green_func_fnames, single_force_green_func_fnames, data_labels,
        inversion_type, perform_normallised_waveform_inversion, compare_all_waveforms_simultaneously, num_samples,
        comparison_metric, manual_indices_time_shift_MT=(), manual_indices_time_shift_SF=(), uid="event",
        stations=(), cut_phase_start_vals=(), cut_phase_length=0, set_pre_time_shift_values_to_zero_switch=True,
        only_save_non_zero_solns_switch=False, return_absolute_similarity_values_switch=False, seed=0, device=0,
        reference_stream=False, nlloc_hyp_filename=None):
    """The reference's ``run`` (:1161-1233) for one set of Green's functions: load the traces,
    least-squares estimate (saved under ``<outdir>/least_squares_result``), Monte Carlo inversion
    on the GPU scaled to the least-squares amplitude, result and best-fit waveforms pickled in the
    reference's layout.  ``uid`` / ``stations`` replace the NonLinLoc file the reference reads them
    from (needs obspy); plotting is left to the reference's plotting script, which reads the
    files written here.  ``nlloc_hyp_filename``: take ``uid`` / ``stations`` from that NonLinLoc file
    instead (``io.get_event_uid_and_station_data_MTFIT_FORMAT_from_nonlinloc_hyp_file``).
    Returns ``(MTs, MTp, MTp_absolute)``.
    """
    from . import io
    if nlloc_hyp_filename:
        uid, stations = io.get_event_uid_and_station_data_MTFIT_FORMAT_from_nonlinloc_hyp_file(nlloc_hyp_filename)
    real, G = io.get_overall_real_and_green_func_data(
        datadir, real_data_fnames, MT_green_func_fnames, single_force_green_func_fnames, inversion_type,
        manual_indices_time_shift_MT, manual_indices_time_shift_SF, cut_phase_start_vals, cut_phase_length,
        set_pre_time_shift_values_to_zero_switch)
    M = perform_inversion(real, G)
    M_amplitude = float(np.sum(M ** 2) ** 0.5)
    lsq_sim = compare_synth_to_real_waveforms(real, G, M, comparison_metric, perform_normallised_waveform_inversion,
                                              compare_all_waveforms_simultaneously, device)
    lsq_dir = os.path.join(outdir, "least_squares_result")
    io.save_to_MTFIT_style_file(M, np.array([lsq_sim]), uid, inversion_type, lsq_dir, stations)
    # the reference slices MTs[:-1] here for the coupled types although the least-squares M has no
    # fraction row; the full M is used instead
    io.save_specific_waveforms_to_file(real, forward_model(G, M, device), data_labels, uid, inversion_type, lsq_dir)
    MTs, MTp, MTp_absolute = perform_monte_carlo_sampled_waveform_inversion(
        real, G, num_samples, M_amplitude, inversion_type, comparison_metric, perform_normallised_waveform_inversion,
        compare_all_waveforms_simultaneously, seed=seed, device=device, reference_stream=reference_stream,
        return_absolute_similarity_values_switch=return_absolute_similarity_values_switch)
    if np.isnan(MTp[0]):
        raise FloatingPointError("sum of probabilities is zero: no adequate solution found (:1208-1210)")
    if only_save_non_zero_solns_switch:
        MTp, MTs = io.remove_zero_prob_results(MTp, MTs)
    io.save_to_MTFIT_style_file(MTs, MTp, uid, inversion_type, outdir, stations, MTp_absolute)
    best = get_synth_forward_model_most_likely_result(MTs, MTp, G, inversion_type, device)
    io.save_specific_waveforms_to_file(real, best, data_labels, uid, inversion_type, outdir)
    return MTs, MTp, MTp_absolute


class MonteCarloPlan:
    """The Green's functions, the data and all work buffers of one inversion problem resident on a
    GPU (``fwi_mc_plan_*``): blocks of up to ``max_samples`` samples are drawn / scored without
    re-uploading or re-allocating anything.  Use it to run more samples than one call can hold::

        with MonteCarloPlan(real_data_array, green_func_array, max_samples=1 << 22) as plan:
            best, total = None, 0.0
            for first in range(0, N, 1 << 22):
                M, frac, sim, like, like_sum = plan.invert("full_mt", min(1 << 22, N - first), seed, first)
                total += like_sum            # posterior of a sample = like / total, once all blocks are in
    """

    def __init__(self, real_data_array, green_func_array, max_samples, device=0):
        G, d = _f64(green_func_array), _f64(real_data_array)
        if G.ndim != 3 or d.shape != (G.shape[0], G.shape[2]):
            raise ValueError("shapes must be G (k, n, t), data (k, t)")
        self.k, self.n, self.t = G.shape
        self.max_samples = int(max_samples)
        self._lib = _lib.load()
        self._plan = C.c_void_p()
        _lib.check(None, self._lib.fwi_mc_plan_create(device, self.k, self.n, self.t, _p(G), _p(d),
                                                      self.max_samples, C.byref(self._plan)))

    def close(self):
        if self._plan is not None and self._plan.value:
            self._lib.fwi_mc_plan_destroy(self._plan)
        self._plan = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _modes(metric, normalise, all_at_once):
        if metric not in _lib.MC_METRICS:
            raise ValueError("comparison_metric must be one of %s" % sorted(_lib.MC_METRICS))
        return _lib.MC_METRICS[metric], int(bool(normalise)), int(bool(all_at_once))

    def invert(self, inversion_type, num_samples, seed=0, first_sample=0, M_amplitude=1.0, comparison_metric="VR",
               perform_normallised_waveform_inversion=True, compare_all_waveforms_simultaneously=True,
               return_samples=True):
        """Draw and score one block on the device: ``(MTs or None, amp_frac or None, similarity,
        likelihood, like_sum)``."""
        if inversion_type not in _lib.MC_INVERSION_TYPES:
            raise ValueError("inversion_type must be one of %s" % sorted(_lib.MC_INVERSION_TYPES))
        N = int(num_samples)
        M = np.empty((self.n, N)) if return_samples else None
        frac = np.empty(N) if return_samples else None
        sim, like = np.empty(N), np.empty(N)
        total, ms = C.c_double(0.0), C.c_double(0.0)
        m, nrm, aao = self._modes(comparison_metric, perform_normallised_waveform_inversion,
                                  compare_all_waveforms_simultaneously)
        _lib.check(None, self._lib.fwi_mc_plan_invert(
            self._plan, _lib.MC_INVERSION_TYPES[inversion_type], int(seed), int(first_sample), N, float(M_amplitude),
            m, nrm, aao, _p(M) if return_samples else None, _p(frac) if return_samples else None, _p(sim), _p(like),
            C.byref(total), C.byref(ms)))
        self.last_kernel_ms = ms.value
        return M, frac, sim, like, total.value

    def score(self, MTs, comparison_metric="VR", perform_normallised_waveform_inversion=True,
              compare_all_waveforms_simultaneously=True):
        """Score one block of given samples ``(n, N)``: ``(similarity, likelihood, like_sum)``."""
        M = _f64(MTs)
        if M.ndim != 2 or M.shape[0] != self.n:
            raise ValueError("MTs must be (n, N)")
        N = M.shape[1]
        sim, like = np.empty(N), np.empty(N)
        total, ms = C.c_double(0.0), C.c_double(0.0)
        m, nrm, aao = self._modes(comparison_metric, perform_normallised_waveform_inversion,
                                  compare_all_waveforms_simultaneously)
        _lib.check(None, self._lib.fwi_mc_plan_score(self._plan, N, _p(M), m, nrm, aao, _p(sim), _p(like),
                                                     C.byref(total), C.byref(ms)))
        self.last_kernel_ms = ms.value
        return sim, like, total.value


def monte_carlo_best_of(real_data_array, green_func_array, num_samples, inversion_type="full_mt", seed=0,
                        M_amplitude=1.0, comparison_metric="VR", perform_normallised_waveform_inversion=True,
                        compare_all_waveforms_simultaneously=True, block=1 << 22, keep=1000, device=0):
    """A run too large to return whole (``num_samples`` up to billions): blocks of ``block`` samples are
    drawn and scored on the device, only the scores come back, and the ``keep`` most likely samples
    are regenerated from ``(seed, index)`` at the end.  Returns ``(indices, MTs (n[+1], keep), MTp
    (keep,), like_total)`` with ``MTp`` normalised over ALL ``num_samples`` (:847-848)."""
    N = int(num_samples)
    block = int(min(block, N))
    best_like, best_idx, total = np.empty(0), np.empty(0, np.int64), 0.0
    with MonteCarloPlan(real_data_array, green_func_array, block, device) as plan:
        for first in range(0, N, block):
            cnt = min(block, N - first)
            _, _, _, like, s = plan.invert(inversion_type, cnt, seed, first, M_amplitude, comparison_metric,
                                           perform_normallised_waveform_inversion,
                                           compare_all_waveforms_simultaneously, return_samples=False)
            total += s
            # candidates: everything above the current keep-th best (one vectorised compare per block;
            # NaN likelihoods compare false and drop out)
            thr = best_like.min() if len(best_like) >= keep else -np.inf
            top = np.flatnonzero(like > thr)
            best_like = np.concatenate([best_like, like[top]])
            best_idx = np.concatenate([best_idx, top.astype(np.int64) + first])
            if len(best_like) > keep:
                sel = np.argpartition(best_like, len(best_like) - keep)[len(best_like) - keep:]
                best_like, best_idx = best_like[sel], best_idx[sel]
    order = np.argsort(-best_like, kind="stable")
    best_like, best_idx = best_like[order], best_idx[order]
    cols, fracs = [], []
    for i in best_idx:  # counter-based sampler: any sample is a pure function of (seed, index)
        M, f = sample_on_device(inversion_type, 1, seed, int(i), M_amplitude, device)
        cols.append(M)
        fracs.append(f)
    MTs = np.hstack(cols)
    if inversion_type in samplers.COUPLED_TYPES:
        MTs = np.vstack((MTs, np.concatenate(fracs)))
    return best_idx, MTs, best_like / total, total
