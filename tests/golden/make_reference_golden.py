"""Golden vectors of the reference's REAL hot loop, produced by the reference itself.

The reference (/root/reference/full_waveform_inversion.py) is Python 2 and imports obspy,
neither of which exists here.  Following SURVEY.md s.8c it is converted IN MEMORY with lib2to3,
given a stub `obspy` module and exec'd into a fresh namespace (its top level only defines
globals and functions).  Nothing of the converted text is written anywhere; only seeded
input / output ARRAYS are saved, as tests/golden/ref_mc_*.npz.

Covered reference functions (file:line): forward_model (:253-264), variance_reduction (:512),
cross_corr_comparison (:534), cross_corr_comparison_shift_allowed (:548),
pearson_correlation_comparison (:568), gaussian_comparison (:578),
compare_synth_to_real_waveforms (:584-684, all 5 metrics x 4 dispatcher modes), the likelihood
map exp(-(1-s)/2) (:774) and the posterior normalisation (:847-848) through
perform_monte_carlo_sampled_waveform_inversion (:786) run with one process and fixed seeds.

Run from the repo root (only where /root/reference exists):
    python tests/golden/make_reference_golden.py
"""
import os
import random
import sys
import types
import warnings

import numpy as np

REF = "/root/reference/full_waveform_inversion.py"
OUT = os.path.dirname(os.path.abspath(__file__))
METRICS = ["VR", "CC", "PCC", "CC-shift", "gau"]


def load_reference():
    from lib2to3 import refactor
    src = open(REF).read()
    tool = refactor.RefactoringTool(refactor.get_fixers_from_package("lib2to3.fixes"))
    py3 = str(tool.refactor_string(src + "\n", "full_waveform_inversion.py"))
    stub = types.ModuleType("obspy")
    stub.UTCDateTime = object
    sys.modules.setdefault("obspy", stub)
    import matplotlib
    matplotlib.use("Agg")
    mod = types.ModuleType("ref_fwi")
    mod.__dict__["__name__"] = "ref_fwi"  # not __main__: the driver at the bottom must not run
    exec(compile(py3, "<reference full_waveform_inversion.py, lib2to3 in memory>", "exec"), mod.__dict__)
    return mod


def make_inputs(seed, k, n, t):
    rng = np.random.default_rng(seed)
    # band-limited "Green's functions": smoothed noise with a decaying envelope
    G = rng.standard_normal((k, n, t))
    ker = np.hanning(9)
    ker /= ker.sum()
    G = np.apply_along_axis(lambda v: np.convolve(v, ker, mode="same"), 2, G)
    G *= np.exp(-np.arange(t) / (0.6 * t))[None, None, :] * (1.0 + rng.random((k, n, 1)))
    M_true = rng.standard_normal((n, 1))
    M_true /= np.linalg.norm(M_true)
    d = np.einsum("kjt,j->kt", G, M_true[:, 0]) + 0.05 * np.abs(G).max() * rng.standard_normal((k, t))
    return G, d, M_true


def case(ref, name, seed, k, n, t, nsamp, inversion_type):
    G, d, M_true = make_inputs(seed, k, n, t)
    np.random.seed(seed)
    random.seed(seed)
    out = {"G": G, "d": d}
    # a batch of samples drawn by the reference's own sampler
    sampler = {"full_mt": ref.generate_random_MT, "single_force": ref.generate_random_single_force_vector,
               "DC": ref.generate_random_DC_MT}[inversion_type]
    Ms = np.hstack([sampler() for _ in range(nsamp - 1)] + [M_true])  # (n, nsamp), last = truth
    out["M"] = Ms
    synth = np.stack([ref.forward_model(G, Ms[:, i:i + 1]) for i in range(nsamp)])
    out["synth"] = synth
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for metric in METRICS:
            for norm in (False, True):
                for allat in (False, True):
                    key = "sim_%s_n%d_a%d" % (metric.replace("-", ""), norm, allat)
                    out[key] = np.array([ref.compare_synth_to_real_waveforms(d, synth[i], metric, norm, allat)
                                         for i in range(nsamp)], dtype=float)
        # the full driver (one process), to pin the likelihood map and the posterior normalisation
        np.random.seed(seed + 100)
        random.seed(seed + 100)
        MTs, MTp, MTp_abs = ref.perform_monte_carlo_sampled_waveform_inversion(
            d, G, num_samples=nsamp, M_amplitude=1.3, inversion_type=inversion_type, comparison_metric="VR",
            perform_normallised_waveform_inversion=False, compare_all_waveforms_simultaneously=False,
            num_processors=1, return_absolute_similarity_values_switch=True)
    out["drv_MTs"], out["drv_MTp"], out["drv_MTp_absolute"] = MTs, MTp, MTp_abs
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "k,n,t =", (k, n, t), "samples", nsamp, "VR(truth, per-trace) =", out["sim_VR_n0_a0"][-1])


if __name__ == "__main__":
    ref = load_reference()
    case(ref, "ref_mc_fullmt", 0, 5, 6, 160, 24, "full_mt")
    case(ref, "ref_mc_force", 1, 21, 3, 100, 16, "single_force")
    case(ref, "ref_mc_dc", 2, 3, 6, 512, 12, "DC")
