"""SURVEY s.8f-4 on the GPU: the reference's similarity measures (full_waveform_inversion.py:512-582) as FWI
misfits, back-propagated through the HIP engine.  For every objective: the misfit and its gradient through
``Engine`` against (a) the same shot loop on the CPU oracle engine and (b) a directional finite difference of
the misfit taken with the GPU engine itself."""
import os
import sys

import numpy as np
import pytest

from full_waveform_inversion_amd import Engine, objectives as ob, shots as sh, workloads

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _oracle_engine import OracleEngine  # noqa: E402

pytestmark = pytest.mark.gpu

CASES = [("l2", {}), ("VR", {"per_trace": True}), ("VR", {"per_trace": False}), ("CC", {"per_trace": True}),
         ("CC", {"per_trace": False}), ("CC-shift", {"per_trace": True}), ("gau", {"per_trace": False}),
         ("gau", {"per_trace": True})]


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b))


def _problem(ndim):
    if ndim == 2:
        w = workloads.cfg3(0.0625, nshots=2)           # 64 x 64, 125 steps, O(8) + sponge
    else:
        w = workloads.cfg5(0.125, nshots=4)            # 32^3, 125 steps, O(8) + sponge
        w.src_idx = w.src_idx[:2]
    wav = w.wavelet(np.float64)
    shots = [sh.Shot(w.src_idx[i:i + 1], wav, w.rec_idx) for i in range(len(w.src_idx))]
    o = OracleEngine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml)
    sh.model_data(o, w.c, shots)  # observed data: the shared input of both paths
    rng = np.random.default_rng(7)
    for s in shots:  # recorded data carry noise: it also gives `gau` the noise level it reads off the trace tails
        s.d_obs = s.d_obs + 0.02 * np.sqrt(np.mean(s.d_obs ** 2)) * rng.standard_normal(s.d_obs.shape)
    return w, shots, o


@pytest.mark.parametrize("ndim", [2, 3])
@pytest.mark.parametrize("name,kw", CASES)
def test_objective_gradient_fp64_engine_vs_oracle_and_finite_difference(gpu, ndim, name, kw):
    w, shots, o = _problem(ndim)
    f = lambda s, d: ob.OBJECTIVES[name](s, d, **kw)  # noqa: E731
    c0 = 0.97 * w.c_init
    J_ref, g_ref = sh.misfit_and_gradient(o, c0, shots, objective=f)
    with Engine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml, sigma_max=o.sigma_max,
                dtype="float64") as e:
        J, g = sh.misfit_and_gradient(e, c0, shots, objective=f)
        assert abs(J - J_ref) < 1e-9 * abs(J_ref) and rel(g, g_ref) < 1e-9
        dc = np.random.default_rng(2).standard_normal(c0.shape)
        eps = 1e-2
        Jp = sh.misfit_and_gradient(e, c0 + eps * dc, shots, objective=f)[0]
        Jm = sh.misfit_and_gradient(e, c0 - eps * dc, shots, objective=f)[0]
    fd = (Jp - Jm) / (2 * eps)
    assert abs(fd - np.sum(g * dc)) < 5e-5 * abs(fd)


@pytest.mark.parametrize("ndim,kernel", [(2, "step2d_fused"), (3, "step3d_stream")])
@pytest.mark.parametrize("name,kw", CASES)
def test_objective_gradient_fp32_engine_vs_oracle(gpu, ndim, kernel, name, kw):
    """The production fp32 kernels.  Every path forms its own adjoint source from its own synthetics, so the
    fp32 forward error (~1e-6 of |d|) enters amplified by |d| / |d_syn - d_obs| (~40 here): 2e-4 on the
    gradient is that product with a margin of 5, not a kernel tolerance."""
    w, shots, o = _problem(ndim)
    f = lambda s, d: ob.OBJECTIVES[name](s, d, **kw)  # noqa: E731
    c0 = 0.97 * w.c_init
    J_ref, g_ref = sh.misfit_and_gradient(o, c0, shots, objective=f)
    with Engine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml, sigma_max=o.sigma_max) as e:
        J, g = sh.misfit_and_gradient(e, c0.astype(np.float32), shots, objective=f)
        assert e.kernel_name == kernel
    print("%s %s %dD: J rel %.2e, g rel %.2e" % (name, kw, ndim, abs(J - J_ref) / abs(J_ref), rel(g, g_ref)))
    assert abs(J - J_ref) < 1e-4 * abs(J_ref) and rel(g, g_ref) < 2e-4
