"""CPU restatement of the reference's REAL hot loop -- TEST INFRASTRUCTURE ONLY.

Unlike the wave-propagation oracle, this one IS pinned: it restates reference code that exists
(/root/reference/full_waveform_inversion.py) and is checked against golden vectors produced by
that code itself (tests/golden/ref_mc_*.npz, made by tests/golden/make_reference_golden.py).
Only tests/ may import it.  Each function cites the reference lines it follows, including the
quirks of SURVEY.md Appendix A (A-4 clamps, A-5 per-trace 'gau' is always 0, A-6 CC-shift rolls
both signals).
"""
import numpy as np


def forward_model(G, M):
    """synth[i, :] = sum_j G[i, j, :] * M[j]   (full_waveform_inversion.py:253-264)."""
    M = np.asarray(M, float).reshape(-1)
    synth = np.zeros(G[:, 0, :].shape, dtype=float)
    for i in range(G.shape[0]):
        for j in range(len(M)):
            synth[i, :] += G[i, j, :] * M[j]
    return synth


def variance_reduction(data, synth):
    """max(0, 1 - sum (d-s)^2 / sum d^2)   (:512-520)."""
    vr = 1.0 - np.sum(np.square(data - synth)) / np.sum(np.square(data))
    return 0.0 if vr < 0.0 else vr


def cross_corr_comparison(data, synth):
    """Zero-lag normalised correlation of data with the standardised synthetic, clamped at 0 (:534-546).

    The reference evaluates it with scipy.signal.correlate(mode='valid', method='fft') on
    equal-length inputs, i.e. one zero-lag product sum.
    """
    sn = (synth - np.mean(synth)) / np.std(synth) / len(synth)
    ncc = np.sum(data * sn) / np.std(data)
    return 0.0 if ncc < 0.0 else ncc


def cross_corr_comparison_shift_allowed(data, synth, max_samples_shift_limit=5):
    """CC on 4x linearly upsampled signals, max over 40 circular shifts applied to BOTH (:548-566)."""
    up = 4
    x = np.arange(0.0, len(data), 1.0 / up)
    dh = np.interp(x, np.arange(len(data)), data)
    sh = np.interp(x, np.arange(len(synth)), synth)
    shifts = np.arange(-max_samples_shift_limit * up, max_samples_shift_limit * up, dtype=int)
    return max(cross_corr_comparison(np.roll(dh, s), np.roll(sh, s)) for s in shifts)


def pearson_correlation_comparison(data, synth):
    """Pearson r clamped at 0 (:568-576)."""
    cov = np.sum((data - np.average(data)) * (synth - np.average(synth))) / len(data)
    pcc = cov / (np.std(data) * np.std(synth))
    return 0.0 if pcc < 0.0 else pcc


def gaussian_comparison(data, synth):
    """exp(-sum (d-s)^2 / (2 sigma^2)), sigma = mean |d[-60:-10]|   (:578-582)."""
    sig = np.average(np.absolute(data[-60:-10]))
    return np.exp(-1 * np.sum(((data - synth) ** 2) / (2 * (sig ** 2))))


_METRICS = {"VR": variance_reduction, "CC": cross_corr_comparison, "PCC": pearson_correlation_comparison,
            "CC-shift": cross_corr_comparison_shift_allowed, "gau": gaussian_comparison}


def compare_synth_to_real_waveforms(real, synth, metric, normalise=True, all_at_once=True):
    """Dispatcher (:584-684): optional per-trace max-abs normalisation of both arrays, then the
    metric on the flattened arrays or per trace with an equal-weight mean."""
    if normalise:  # :595-599 / :639-643
        real = real / np.max(np.absolute(real), axis=1, keepdims=True)
        synth = synth / np.max(np.absolute(synth), axis=1, keepdims=True)
    f = _METRICS[metric]
    if all_at_once:
        return f(real.flatten(), synth.flatten())
    if metric == "gau":
        # :661 / :680 assign the per-trace value to the wrong variable; the mean of the untouched
        # zero array is returned (verified by running the reference)
        return 0.0
    return float(np.average([f(real[k, :], synth[k, :]) for k in range(real.shape[0])]))


def likelihood(similarity):
    """exp(-(1 - s) / 2)   (:774)."""
    return np.exp(-(1.0 - np.asarray(similarity, float)) / 2.0)


def posterior(like):
    """MTp = L p_model / sum(p_model L)   (:847-848); p_model = 1/N cancels."""
    like = np.asarray(like, float)
    p_model = 1.0 / len(like)
    return like * p_model / np.sum(p_model * like)


def score_samples(G, d, Ms, metric="VR", normalise=False, all_at_once=False):
    """Steps 4-7 of the worker loop (:713-774) for given samples Ms (n, N): similarity, likelihood."""
    sims = np.array([compare_synth_to_real_waveforms(d, forward_model(G, Ms[:, i]), metric, normalise,
                                                     all_at_once) for i in range(Ms.shape[1])], float)
    return sims, likelihood(sims)
