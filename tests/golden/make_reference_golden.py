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

ref_samplers.npz pins the seven samplers (:282-510) and the driver's handling of every
inversion_type (:740-760, :852-853): after np.random.seed(s); random.seed(s) each sampler is
called 48 times and its outputs kept, and the one-process driver is run per type on a small
seeded problem.

ref_pipeline.npz pins the host functions either side of the loop: load_input_data (:75-113),
get_overall_real_and_green_func_data (:168-197), perform_inversion (:242-251) and
get_synth_forward_model_most_likely_result (:974-1020), run on text trace files written to a
temporary directory from the seeded arrays stored alongside the outputs.

ref_multimedia_pipeline.npz pins the two-media host functions: load_input_data_multiple_media (:116-165), the
two-media switch of get_overall_real_and_green_func_data and the two-media branches of
get_synth_forward_model_most_likely_result.

ref_names.npz pins the small module-level functions `reference_api` re-exposes under the reference's names: the tensor
helpers (:199-241), the similarity measures on one pair of 1-D arrays (:512-582) and the dispatcher on a precomputed
synthetic (:584-684).

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


SAMPLERS = {"full_mt": "generate_random_MT", "DC": "generate_random_DC_MT",
            "single_force": "generate_random_single_force_vector",
            "DC_single_force_couple": "generate_random_DC_single_force_coupled_tensor",
            "DC_single_force_no_coupling": "generate_random_DC_single_force_uncoupled_tensor",
            "DC_crack_couple": "generate_random_DC_crack_coupled_tensor",
            "single_force_crack_no_coupling": "generate_random_single_force_crack_uncoupled_tensor"}


class _InProcess:
    """Stand-in for the `multiprocessing` module inside the reference's driver: runs the worker in
    THIS process.  Needed for the coupled types only -- Python 3 re-seeds the stdlib `random` module
    in every forked child (os.register_at_fork), which the reference's Python 2 did not, so a forked
    worker's amplitude fractions would not be reproducible from the seed.  The code under test
    (driver, worker, samplers, metrics) is still the reference's."""

    class Process:
        def __init__(self, target, args):
            self.target, self.args = target, args

        def start(self):
            self.target(*self.args)

        def join(self):
            pass

    class _Manager:
        @staticmethod
        def dict():
            return {}

    @staticmethod
    def Manager():
        return _InProcess._Manager()


def samplers_case(ref, name, nsamp=48, ndrv=40):
    out = {}
    ref.multiprocessing = _InProcess
    for idx, (typ, fn) in enumerate(SAMPLERS.items()):
        seed = 1000 + idx
        np.random.seed(seed)
        random.seed(seed)
        cols, fracs = [], []
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for _ in range(nsamp):
                r = getattr(ref, fn)()
                if isinstance(r, tuple):
                    cols.append(r[0])
                    fracs.append(r[1])
                else:
                    cols.append(r)
        out["seed_" + typ] = seed
        out["M_" + typ] = np.hstack(cols)
        if fracs:
            out["frac_" + typ] = np.array(fracs)
        # the driver for this type (one process): MTs incl. the amplitude-fraction row, posterior, likelihoods
        n = out["M_" + typ].shape[0]
        G, d, _ = make_inputs(50 + idx, 4, n, 64)
        np.random.seed(seed + 7)
        random.seed(seed + 7)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            MTs, MTp, MTp_abs = ref.perform_monte_carlo_sampled_waveform_inversion(
                d, G, num_samples=ndrv, M_amplitude=0.7, inversion_type=typ, comparison_metric="PCC",
                perform_normallised_waveform_inversion=True, compare_all_waveforms_simultaneously=False,
                num_processors=1, return_absolute_similarity_values_switch=True)
        out["drv_G_" + typ], out["drv_d_" + typ] = G, d
        out["drv_MTs_" + typ], out["drv_MTp_" + typ], out["drv_MTp_absolute_" + typ] = MTs, MTp, MTp_abs
        print(typ, "samples", out["M_" + typ].shape, "driver MTs", MTs.shape)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def multimedia_case(ref, name, nsamp=24):
    """The worker's two-media branch with one fraction per phase type (:715-727; the one-fraction branch, :730-731,
    cannot run past its first sample: it overwrites its own Green's functions): Green's
    functions (k, n, t, 2), phase labels per trace, the driver's three extra fraction rows (:855-861)."""
    ref.multiprocessing = _InProcess
    out = {}
    for typ, n in (("full_mt", 6), ("DC_single_force_couple", 9)):
        k, t = 5, 48
        G0, d, _ = make_inputs(300 + n, k, n, t)
        G1 = make_inputs(400 + n, k, n, t)[0]
        G2 = np.stack((G0, 0.7 * G0 + 0.6 * G1), axis=3)
        labels = ["P", "S", "surface", "P", "S"]
        seed = 9000 + n
        np.random.seed(seed)
        random.seed(seed)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            MTs, MTp, MTp_abs = ref.perform_monte_carlo_sampled_waveform_inversion(
                d, G2, num_samples=nsamp, M_amplitude=0.9, inversion_type=typ, comparison_metric="VR",
                perform_normallised_waveform_inversion=False, compare_all_waveforms_simultaneously=False,
                num_processors=1, return_absolute_similarity_values_switch=True,
                invert_for_ratio_of_multiple_media_greens_func_switch=True, green_func_phase_labels=labels,
                num_phase_types_for_media_ratios=3)
        out["seed_" + typ], out["G2_" + typ], out["d_" + typ] = seed, G2, d
        out["labels"] = np.array(labels)
        out["MTs_" + typ], out["MTp_" + typ], out["MTp_absolute_" + typ] = MTs, MTp, MTp_abs
        print("multi-media", typ, "MTs", MTs.shape)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def write_trace_files(datadir, real, mt, sf):
    """One text file per trace, in the layout load_input_data (:75-113) reads: data = t values,
    Green's functions = t rows x n component columns.  Returns the three file-name lists."""
    names = ([], [], [])
    for i in range(real.shape[0]):
        for lst, stem, arr in ((names[0], "real_%02d.txt", real[i]), (names[1], "gf_mt_%02d.txt", mt[i].T),
                               (names[2], "gf_sf_%02d.txt", sf[i].T)):
            lst.append(stem % i)
            np.savetxt(os.path.join(datadir, stem % i), arr, fmt="%.18e")
    return names


def pipeline_case(ref, name):
    """The host functions either side of the loop: trace loading with time shift / phase cut and the
    unit scalings (:75-113, :168-197), the least-squares estimate (:242-251) and the best-sample
    synthetic (:974-1020)."""
    import tempfile
    rng = np.random.default_rng(77)
    k, t = 5, 120
    real = rng.standard_normal((k, t))
    mt = rng.standard_normal((k, 6, t)) * 1e-10
    sf = rng.standard_normal((k, 3, t)) * 1e-7
    out = {"real": real, "mt": mt, "sf": sf}
    shift_mt, shift_sf = [3, 0, 7, 1, 12], [2, 5, 0, 9, 4]
    cut_start, cut_len = [10, 0, 33, 20, 58], 60
    out["shift_mt"], out["shift_sf"], out["cut_start"], out["cut_len"] = shift_mt, shift_sf, cut_start, cut_len
    with tempfile.TemporaryDirectory() as tmp:
        rn, mn, sn = write_trace_files(tmp, real, mt, sf)
        out["plain_real"], out["plain_green"] = ref.load_input_data(tmp, rn, mn)
        out["shift_real"], out["shift_green"] = ref.load_input_data(tmp, rn, mn, shift_mt)
        out["shiftkeep_real"], out["shiftkeep_green"] = ref.load_input_data(
            tmp, rn, mn, shift_mt, set_pre_time_shift_values_to_zero_switch=False)
        out["cut_real"], out["cut_green"] = ref.load_input_data(tmp, rn, sn, shift_sf, cut_start, cut_len)
        for typ in SAMPLERS:
            r, g = ref.get_overall_real_and_green_func_data(
                tmp, rn, mn, sn, typ, manual_indices_time_shift_MT=shift_mt, manual_indices_time_shift_SF=shift_sf,
                cut_phase_start_vals=cut_start, cut_phase_length=cut_len)
            out["overall_real_" + typ], out["overall_green_" + typ] = r, g
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                out["lsq_" + typ] = ref.perform_inversion(r, g)
            # most likely synthetic from a small set of samples with a known best one
            np.random.seed(5)
            random.seed(5)
            cols, fr = [], []
            for _ in range(6):
                smp = getattr(ref, SAMPLERS[typ])()
                cols.append(smp[0] if isinstance(smp, tuple) else smp)
                fr.append(smp[1] if isinstance(smp, tuple) else 0.0)
            MTs = np.hstack(cols)
            if isinstance(smp, tuple):
                MTs = np.vstack((MTs, np.array(fr)))
            MTp = np.array([0.1, 0.05, 0.4, 0.2, 0.15, 0.1])
            out["ml_MTs_" + typ], out["ml_MTp_" + typ] = MTs, MTp
            out["ml_synth_" + typ] = ref.get_synth_forward_model_most_likely_result(MTs, MTp, g, typ)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "written:", len(out), "arrays")


def multimedia_pipeline_case(ref, name):
    """The two-media host functions: load_input_data_multiple_media (:116-165), the switch of
    get_overall_real_and_green_func_data (:168-197) and the two-media branches of
    get_synth_forward_model_most_likely_result (:974-1020; one fraction, and one per phase type, for a coupled and a
    plain inversion type).  File lists hold medium 1's files followed by medium 2's."""
    import tempfile
    rng = np.random.default_rng(78)
    k, t = 5, 64
    real = rng.standard_normal((k, t))
    mt = rng.standard_normal((2, k, 6, t)) * 1e-10   # [medium]
    sf = rng.standard_normal((2, k, 3, t)) * 1e-7
    shift_mt, shift_sf = [3, 0, 7, 1, 12], [2, 5, 0, 9, 4]
    cut_start, cut_len = [10, 0, 23, 20, 28], 36
    labels = ["P", "S", "surface", "S", "P"]
    out = {"real": real, "mt": mt, "sf": sf, "shift_mt": shift_mt, "shift_sf": shift_sf, "cut_start": cut_start,
           "cut_len": cut_len, "labels": np.array(labels)}
    with tempfile.TemporaryDirectory() as tmp:
        rn, mn, sn = [], [], []
        for m in (0, 1):
            r_, m_, s_ = write_trace_files(tmp, real, mt[m], sf[m])
            # (write_trace_files names by trace only: give each medium its own names)
            for lst, src in ((mn, m_), (sn, s_)):
                for f in src:
                    g = "m%d_%s" % (m + 1, f)
                    os.replace(os.path.join(tmp, f), os.path.join(tmp, g))
                    lst.append(g)
            rn = r_
        out["plain_real"], out["plain_green"] = ref.load_input_data_multiple_media(tmp, rn, mn, k)
        out["shift_real"], out["shift_green"] = ref.load_input_data_multiple_media(tmp, rn, mn, k, shift_mt)
        out["shiftkeep_real"], out["shiftkeep_green"] = ref.load_input_data_multiple_media(
            tmp, rn, mn, k, shift_mt, set_pre_time_shift_values_to_zero_switch=False)
        out["cut_real"], out["cut_green"] = ref.load_input_data_multiple_media(tmp, rn, sn, k, shift_sf, cut_start,
                                                                               cut_len)
        for typ in SAMPLERS:
            r, g = ref.get_overall_real_and_green_func_data(
                tmp, rn, mn, sn, typ, manual_indices_time_shift_MT=shift_mt, manual_indices_time_shift_SF=shift_sf,
                cut_phase_start_vals=cut_start, cut_phase_length=cut_len,
                invert_for_ratio_of_multiple_media_greens_func_switch=True, green_func_fnames_split_index=k)
            out["overall_real_" + typ], out["overall_green_" + typ] = r, g
            # most likely synthetic: samples with fraction rows appended the way the driver does (:855-864)
            np.random.seed(6)
            random.seed(6)
            cols, fr = [], []
            for _ in range(6):
                smp = getattr(ref, SAMPLERS[typ])()
                cols.append(smp[0] if isinstance(smp, tuple) else smp)
                fr.append(smp[1] if isinstance(smp, tuple) else 0.0)
            base = np.hstack(cols)
            if isinstance(smp, tuple):
                base = np.vstack((base, np.array(fr)))
            MTp = np.array([0.1, 0.05, 0.15, 0.2, 0.4, 0.1])
            for nphase, nfr in ((0, 1), (3, 3)):
                MTs = np.vstack((base, np.random.uniform(0.0, 1.0, (nfr, 6))))
                out["ml_MTs_%d_%s" % (nphase, typ)] = MTs
                out["ml_synth_%d_%s" % (nphase, typ)] = ref.get_synth_forward_model_most_likely_result(
                    MTs, MTp, g, typ, invert_for_ratio_of_multiple_media_greens_func_switch=True,
                    green_func_phase_labels=labels, num_phase_types_for_media_ratios=nphase)
            out["ml_MTp"] = MTp
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "written:", len(out), "arrays")


def names_case(ref, name):
    """Inputs / outputs of the reference's small module-level functions, for `reference_api` (same names, same positional
    signatures): the tensor helpers (:199-241), the similarity measures on ONE pair of 1-D arrays (:512-582, incl. the
    unused variance_reduction_normallised) and the dispatcher on a precomputed synthetic (:584-684)."""
    rng = np.random.default_rng(91)
    out = {}
    mt = rng.standard_normal(6)
    full = ref.get_full_MT_array(mt)
    theta, phi = 0.7, -2.1
    force = rng.standard_normal(3)
    out.update(mt=mt, full=full, theta=theta, phi=phi, force=force, six_back=ref.get_six_MT_from_full_MT_array(full),
               eig=np.array(ref.find_eigenvalues_from_sixMT(mt)), rot_mt=ref.rot_mt_by_theta_phi(full, theta, phi),
               rot_mt_default=ref.rot_mt_by_theta_phi(full), rot_force=ref.rot_single_force_by_theta_phi(force, theta, phi))
    G, d, M_true = make_inputs(17, 4, 6, 128)
    synth = ref.forward_model(G, M_true + 0.3 * rng.standard_normal((6, 1)))
    out.update(d=d, synth=synth)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for fn in ("variance_reduction", "variance_reduction_normallised", "cross_corr_comparison",
                   "cross_corr_comparison_shift_allowed", "pearson_correlation_comparison", "gaussian_comparison"):
            out["pair_" + fn] = np.array([getattr(ref, fn)(d[i], synth[i]) for i in range(d.shape[0])])
        for metric in METRICS:
            for norm in (False, True):
                for allat in (False, True):
                    out["cmp_%s_%d_%d" % (metric, norm, allat)] = ref.compare_synth_to_real_waveforms(
                        d, synth, metric, norm, allat)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "written:", len(out), "arrays")


if __name__ == "__main__":
    ref = load_reference()
    if len(sys.argv) > 1:  # only the named cases, e.g. `... make_reference_golden.py multimedia_pipeline`
        for nm in sys.argv[1:]:
            globals()[nm + "_case"](ref, "ref_" + nm)
        sys.exit(0)
    pipeline_case(ref, "ref_pipeline")
    multimedia_pipeline_case(ref, "ref_multimedia_pipeline")
    names_case(ref, "ref_names")
    samplers_case(ref, "ref_samplers")
    multimedia_case(ref, "ref_multimedia")
    case(ref, "ref_mc_fullmt", 0, 5, 6, 160, 24, "full_mt")
    case(ref, "ref_mc_force", 1, 21, 3, 100, 16, "single_force")
    case(ref, "ref_mc_dc", 2, 3, 6, 512, 12, "DC")
