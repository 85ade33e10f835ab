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


def mixed_media_greens(G2, frac_by_phase, labels):
    """Per-sample Green's functions of the two-media branch with one fraction per phase type (:715-727):
    trace j takes (1 - f) G[j, :, :, 0] + f G[j, :, :, 1] with f the fraction of its phase label."""
    G = np.zeros(G2[:, :, :, 0].shape, dtype=float)
    for j in range(len(labels)):
        f = frac_by_phase[labels[j]]
        G[j, :, :] = (1.0 - f) * G2[j, :, :, 0] + f * G2[j, :, :, 1]
    return G


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


# ---------------------------------------------------------------------------------------------
# The device sampler's deviate generator (csrc/mc_kernels.hip, mc_sample_kernel).  BUILD-DEFINED:
# the reference draws from numpy's / the stdlib's global Mersenne Twisters, whose sequential
# streams cannot be split over GPU threads; the device uses a counter-based generator instead.
# Philox4x32-10 is the published algorithm of Salmon et al., "Parallel random numbers: as easy as
# 1, 2, 3" (SC'11); PHILOX_KAT are two known-answer vectors from its Random123 distribution
# (kat_vectors: the all-zero input and the digits-of-pi input), checked in tests/test_mc_oracle.py.
# The deterministic maps from deviates to samples are the reference's and are pinned separately
# (tests/test_samplers.py).
# ---------------------------------------------------------------------------------------------
PHILOX_KAT = [  # (counter[4], key[2]) -> output[4]
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32 with 10 rounds on uint32 arrays (broadcast together); returns 4 uint32 arrays."""
    u64 = np.uint64
    c0, c1, c2, c3 = [np.asarray(c, dtype=u64) & u64(0xFFFFFFFF) for c in np.broadcast_arrays(c0, c1, c2, c3)]
    k0, k1 = u64(int(k0) & 0xFFFFFFFF), u64(int(k1) & 0xFFFFFFFF)
    mask, sh = u64(0xFFFFFFFF), u64(32)
    for _ in range(10):
        p0, p1 = u64(0xD2511F53) * c0, u64(0xCD9E8D57) * c2
        c0, c1, c2, c3 = (p1 >> sh) ^ c1 ^ k0, p1 & mask, (p0 >> sh) ^ c3 ^ k1, p0 & mask
        k0, k1 = (k0 + u64(0x9E3779B9)) & mask, (k1 + u64(0xBB67AE85)) & mask
    return c0, c1, c2, c3


def _device_uniform_pair(idx, seed, block):
    """Two doubles in [0, 1) of sample indices ``idx`` from Philox block ``block``."""
    idx = np.asarray(idx, dtype=np.uint64)
    r = philox4x32_10(idx & np.uint64(0xFFFFFFFF), idx >> np.uint64(32), np.uint64(block), np.uint64(0),
                      int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF)
    r = [x.astype(np.float64) for x in (r[0] >> np.uint64(5), r[1] >> np.uint64(6), r[2] >> np.uint64(5),
                                        r[3] >> np.uint64(6))]
    scale = 1.0 / 9007199254740992.0
    return (r[0] * 67108864.0 + r[1]) * scale, (r[2] * 67108864.0 + r[3]) * scale


def _device_normal_pair(idx, seed, block):
    ua, ub = _device_uniform_pair(idx, seed, block)
    r, ang = np.sqrt(-2.0 * np.log(1.0 - ua)), 6.283185307179586476925 * ub
    return r * np.cos(ang), r * np.sin(ang)


def device_sampler_deviates(inversion_type, seed, first_sample, num_samples):
    """The deviates mc_sample_kernel feeds the reference's sampler maps, keyed like
    ``full_waveform_inversion_amd.samplers.draw_deviates``: blocks 0..2 are normal pairs
    (z0..z5), blocks 8 and 9 the uniforms (U0, U1), (U2, U3)."""
    idx = np.arange(first_sample, first_sample + num_samples, dtype=np.uint64)
    z = np.stack([v for b in range(3) for v in _device_normal_pair(idx, seed, b)], axis=1)  # (N, 6)
    U0, U1 = _device_uniform_pair(idx, seed, 8)
    U2, U3 = _device_uniform_pair(idx, seed, 9)
    if inversion_type == "full_mt":
        return {"z6": z}
    if inversion_type in ("DC", "single_force"):
        return {"z3": z[:, :3]}
    if inversion_type == "DC_single_force_couple":
        return {"z3": z[:, :3], "frac": U0}
    if inversion_type == "DC_single_force_no_coupling":
        return {"z3_dc": z[:, :3], "z3_sf": z[:, 3:], "frac": U0}
    crack = {"u_theta": 2.0 * U0 - 1.0, "r_phi": U1, "r_quadrant": U2, "frac": U3}
    if inversion_type == "DC_crack_couple":
        crack["z3"] = z[:, :3]
        return crack
    if inversion_type == "single_force_crack_no_coupling":
        crack["z3_sf"], crack["z3_rot"] = z[:, :3], z[:, 3:]
        return crack
    raise ValueError(inversion_type)
