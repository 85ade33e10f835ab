"""Validation ladder of the CPU oracle (oracle/), which is BUILD-DEFINED.

PARITY UNPINNED: the reference has no wave-propagation / adjoint / gradient
code, hence no golden vectors to pin the oracle with.  What
pins it instead: analytic point-source solutions, the exact adjoint identity,
finite-difference gradient checks, stability at the CFL limit, and frozen
fixtures of its own output (tests/golden/, made by tests/golden/make_golden.py)
so that later edits cannot silently change the definition.
"""
import os

import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from oracle import fwi_oracle as fo
from oracle.c_oracle import CPropagator

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b)


def _random_case(shape, order, nt=40, npml=5, seed=0, nsrc=2, nrec=6):
    rng = np.random.default_rng(seed)
    c = 1500.0 + 1000.0 * rng.random(shape)
    h = 10.0
    dt = 0.6 * fo.cfl_dt(c.max(), h, len(shape), order)
    src = np.stack([rng.integers(0, s, nsrc) for s in shape], 1)
    rec = np.stack([rng.integers(0, s, nrec) for s in shape], 1)
    w = rng.standard_normal((nt, nsrc))
    r = rng.standard_normal((nt, nrec))
    return c, h, dt, src, rec, w, r


CASES = [((30, 26), 2), ((30, 26), 4), ((30, 26), 8), ((14, 12, 16), 2), ((14, 12, 16), 8)]


@pytest.mark.parametrize("shape,order", CASES)
def test_adjoint_identity(shape, order):
    """<F s, r> = <s, F^T r> to round-off: the adjoint is the exact transpose."""
    c, h, dt, src, rec, w, r = _random_case(shape, order)
    p = fo.Propagator(c, h, dt, order, npml=5)
    d = p.forward(src, w, rec)
    a = p.adjoint(r, image=False)
    lhs, rhs = np.sum(d * r), np.sum(w * a)
    assert abs(lhs - rhs) <= 1e-12 * max(abs(lhs), abs(rhs))


@pytest.mark.parametrize("shape,order", [((30, 26), 8), ((30, 26), 2), ((14, 12, 16), 8)])
@pytest.mark.parametrize("wrt", ["velocity", "slowness2"])
def test_gradient_matches_finite_differences(shape, order, wrt):
    c, h, dt, src, rec, w, _ = _random_case(shape, order, seed=3)
    p0 = fo.Propagator(c, h, dt, order, npml=5)
    d_obs = p0.forward(src, w, rec, save=False)
    rng = np.random.default_rng(7)
    c1 = c * (1.0 + 0.02 * rng.standard_normal(shape))
    dc = rng.standard_normal(shape)
    sm = p0.sigma_max

    def J(cc):
        return fo.misfit_and_gradient(cc, h, dt, order, 5, src, w, rec, d_obs, sigma_max=sm, wrt=wrt)

    _, g, _ = J(c1)
    eps = 1e-3
    if wrt == "velocity":
        fd = (J(c1 + eps * dc)[0] - J(c1 - eps * dc)[0]) / (2 * eps)
    else:  # perturb m = 1/c^2
        m1 = 1.0 / c1 ** 2
        dm = dc * 1e-9
        fd = (J((m1 + eps * dm) ** -0.5)[0] - J((m1 - eps * dm) ** -0.5)[0]) / (2 * eps)
        dc = dm
    an = np.sum(g * dc)
    assert abs(fd - an) <= 1e-6 * abs(fd)


def _ricker_at(t, f0):
    """The wavelet of fo.ricker() at arbitrary times (closed form: no interpolation of its samples)."""
    a = (np.pi * f0 * (t - 1.5 / f0)) ** 2
    return (1.0 - 2.0 * a) * np.exp(-a)


def _analytic_3d(fac, oracle):
    """Homogeneous 3-D: u(r, t) = w(t - r/c) / (4 pi r) for (1/c^2) u_tt - lap u = w delta.  Returns the relative L2
    errors at two offsets for the time step fac x CFL over a fixed physical time."""
    n, h, c0, f0 = 72, 10.0, 2000.0, 12.0
    T = 190 * 0.4 * fo.cfl_dt(c0, h, 3, 8)
    dt = fac * fo.cfl_dt(c0, h, 3, 8)
    nt = int(round(T / dt))
    w = _ricker_at(np.arange(nt) * dt, f0)
    offs = np.array([12, 18])
    rec = np.stack([np.full(2, n // 2), np.full(2, n // 2), n // 2 + offs], 1)
    d = oracle(np.full((n, n, n), c0), h, dt, 8, 0).forward([[n // 2] * 3], w, rec, save=False)
    t = (np.arange(nt) + 1) * dt  # d[n] samples u^{n+1}
    return [rel(d[:, i], _ricker_at(t - o * h / c0, f0) * (t > o * h / c0) / (4 * np.pi * o * h))
            for i, o in enumerate(offs)]


def _analytic_2d(fac, oracle):
    """Homogeneous 2-D: u = (1/2pi) int w(tau) H(t-tau-r/c) / sqrt((t-tau)^2 - r^2/c^2) dtau."""
    n, h, c0, f0, off = 220, 10.0, 2000.0, 10.0, 40
    T = 360 * 0.4 * fo.cfl_dt(c0, h, 2, 8)
    dt = fac * fo.cfl_dt(c0, h, 2, 8)
    nt = int(round(T / dt))
    w = _ricker_at(np.arange(nt) * dt, f0)
    d = oracle(np.full((n, n), c0), h, dt, 8, 0).forward([[n // 2, n // 2]], w, [[n // 2, n // 2 + off]], save=False)[:, 0]
    r = off * h
    t = (np.arange(nt) + 1) * dt
    # substitution t - tau = (r/c) cosh(s) removes the inverse-square-root singularity
    s = np.linspace(0.0, 7.0, 20001)
    tau = t[:, None] - (r / c0) * np.cosh(s)[None, :]
    ex = np.trapezoid(np.where(tau > 0, _ricker_at(tau, f0), 0.0), s, axis=1) / (2 * np.pi)
    return [rel(d, ex)]


def test_analytic_point_source_numpy_oracle():
    """The NumPy oracle (the definition) against the closed-form solutions at the coarse time step: a few 1e-3, all of
    it the second-order time discretisation (the refinement below shows that)."""
    assert max(_analytic_3d(0.4, fo.Propagator)) < 4e-3
    assert max(_analytic_2d(0.4, fo.Propagator)) < 6e-3


@pytest.mark.parametrize("case", [_analytic_3d, _analytic_2d])
def test_analytic_point_source_converges_at_second_order_to_1e3(case):
    """the only thing that stands behind the unpinned oracle is its own ladder, so the analytic
    check is a CONVERGENCE statement now, not a 3 % tolerance: halving the time step twice (fixed grid, 16 points per
    wavelength: the O(8) spatial error is ~3e-5) divides the error by 4 each time and lands below 1e-3 / 3e-4.  (The C
    port does the refinement -- it equals the NumPy oracle to 1e-13, test_c_oracle_matches_numpy_oracle.)"""
    from oracle.c_oracle import CPropagator
    e = [case(f, CPropagator) for f in (0.4, 0.2, 0.1)]
    for i in range(len(e[0])):
        assert 3.5 < e[0][i] / e[1][i] < 4.5 and 3.0 < e[1][i] / e[2][i] < 4.5, e   # observed order 2 (the last
        # halving already feels the spatial floor at the nearer 3-D offsets)
        assert e[1][i] < 1.2e-3 and e[2][i] < 3e-4, e


def test_cfl_limit():
    """Stable just below the CFL bound of cfl_dt(), unstable above it."""
    n = 40
    c = np.full((n, n), 2000.0)
    rng = np.random.default_rng(0)
    w = rng.standard_normal((300, 1))
    for fac, stable in [(0.99, True), (1.05, False)]:
        dt = fac * fo.cfl_dt(2000.0, 10.0, 2, 8)
        p = fo.Propagator(c, 10.0, dt, 8, npml=0)
        with np.errstate(over="ignore", invalid="ignore"):
            d = p.forward([[n // 2, n // 2]], w, [[3, 3]], save=False)
        big = not np.all(np.isfinite(d)) or np.abs(d).max() > 1e6 * np.abs(d[:50]).max()
        assert big != stable


def test_sponge_absorbs():
    """With the absorbing border the late-time energy at a receiver is far below the undamped run."""
    n = 90
    c = np.full((n, n), 2000.0)
    dt = 0.7 * fo.cfl_dt(2000.0, 10.0, 2, 8)
    nt = 700
    w = fo.ricker(nt, dt, 15.0)
    late = slice(450, nt)
    e = []
    for npml in (0, 20):
        p = fo.Propagator(c, 10.0, dt, 8, npml=npml)
        d = p.forward([[n // 2, n // 2]], w, [[n // 2, n // 2 + 10]], save=False)
        e.append(np.sum(d[late] ** 2))
    assert e[1] < 1e-3 * e[0]


@pytest.mark.parametrize("shape,order", CASES)
def test_c_oracle_matches_numpy_oracle(shape, order):
    c, h, dt, src, rec, w, r = _random_case(shape, order, seed=5)
    p, q = fo.Propagator(c, h, dt, order, 5), CPropagator(c, h, dt, order, 5)
    assert rel(q.forward(src, w, rec), p.forward(src, w, rec)) < 1e-13
    assert rel(q.q_store, p.q_store) < 1e-13
    assert rel(q.adjoint(r), p.adjoint(r)) < 1e-13
    assert rel(q.gradient(), p.gradient()) < 1e-12


@pytest.mark.parametrize("stride", [2, 3, 5])
def test_image_stride_definition_and_accuracy(stride):
    """image_stride S: img = S * sum_{n % S == 0} mu^{n+1} q^n.  Both oracles agree on it, and for a wavelet
    that S * dt still samples well it approximates the exact discrete gradient."""
    n, order, npml = 40, 8, 4
    c = np.full((n, n), 2000.0)
    c[n // 2:, :] = 2600.0
    h = 10.0
    dt = 0.5 * fo.cfl_dt(c.max(), h, 2, order)   # f0 = 12 Hz, dt ~ 1.2 ms: >= 35 samples per period
    nt = 180
    w = fo.ricker(nt, dt, 12.0)
    src, rec = [[6, 20]], [[6, x] for x in range(4, 36, 4)]
    full = fo.Propagator(c, h, dt, order, npml)
    d = full.forward(src, w, rec)
    r = 0.3 * d
    full.adjoint(r)
    g_full = full.gradient()
    ps = fo.Propagator(c, h, dt, order, npml, image_stride=stride)
    cs = CPropagator(c, h, dt, order, npml, image_stride=stride)
    for p in (ps, cs):
        p.forward(src, w, rec)
        p.adjoint(r)
    g_s = ps.gradient()
    assert rel(cs.gradient(), g_s) < 1e-12
    assert rel(g_s, g_full) < 0.02 * stride   # quadrature error of the Riemann sum, small while oversampled
    assert rel(g_s, g_full) > 1e-9            # ... but it IS a different (decimated) sum


def test_empty_point_sets():
    """No sources -> zero data; no receivers -> (nt, 0) output; both oracles agree."""
    c = np.full((20, 20), 2000.0)
    dt = 0.5 * fo.cfl_dt(2000.0, 10.0, 2, 8)
    for P in (fo.Propagator, CPropagator):
        p = P(c, 10.0, dt, 8, 4)
        d = p.forward(np.zeros((0, 2), int), np.zeros((10, 0)), [[5, 5]])
        assert d.shape == (10, 1) and not d.any()
        d = p.forward([[5, 5]], np.ones((10, 1)), np.zeros((0, 2), int))
        assert d.shape == (10, 0)


def test_index_outside_grid_is_rejected():
    p = fo.Propagator(np.full((10, 10), 2000.0), 10.0, 1e-3, 2)
    with pytest.raises(ValueError):
        p.forward([[10, 0]], np.ones((4, 1)), [[1, 1]])


@pytest.mark.parametrize("name", ["g2d_o8", "g2d_o2", "g3d_o8", "g3d_o4"])
def test_golden_fixtures(name):
    """The oracle still reproduces its own frozen outputs (definition drift guard)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    p = fo.Propagator(z["c"], float(z["h"]), float(z["dt"]), int(z["order"]), int(z["npml"]),
                      sigma_max=float(z["sigma_max"]))
    d = p.forward(z["src_idx"], z["wavelet"], z["rec_idx"])
    assert rel(d, z["seis"]) < 1e-12
    a = p.adjoint(z["residual"])
    assert rel(a, z["adj_src"]) < 1e-12
    assert rel(p.gradient(), z["grad_c"]) < 1e-11


def test_observed_convergence_orders():
    """SURVEY s.4 tier 1: error vs the analytic 3-D solution falls with order ~8 in space when dt is
    small enough, and with order ~2 in time at fixed fine space."""
    c0, f0 = 2000.0, 8.0
    L = 720.0            # physical box edge (m); source in the middle, receiver 180 m away
    T = 0.30

    def err(n, dt_frac, order=8):
        h = L / n
        dt = dt_frac * fo.cfl_dt(c0, h, 3, order)
        nt = int(round(T / dt))
        dt = T / nt
        w = fo.ricker(nt, dt, f0)
        mid = n // 2
        off = int(round(180.0 / h))
        p = fo.Propagator(np.full((n, n, n), c0), h, dt, order, npml=0)
        d = p.forward([[mid, mid, mid]], w, [[mid, mid, mid + off]], save=False)[:, 0]
        t = (np.arange(nt) + 1) * dt
        r = off * h
        # fine quadrature-free reference: the wavelet is analytic
        a = (np.pi * f0 * (t - r / c0 - 1.5 / f0)) ** 2
        ex = (1.0 - 2.0 * a) * np.exp(-a) / (4 * np.pi * r) * (t >= r / c0)
        return np.linalg.norm(d - ex) / np.linalg.norm(ex)

    # space: halve h at a tiny time step (time error negligible): order 8 => ratio >> 2^4
    e_coarse, e_fine = err(24, 0.05), err(48, 0.05)
    assert e_fine < e_coarse / 2 ** 4, (e_coarse, e_fine)
    # time: fine space, halve dt: order 2 => ratio ~ 4
    e1, e2 = err(48, 0.8), err(48, 0.4)
    assert 2.5 < e1 / e2 < 6.0, (e1, e2)


# ---------------------------------------------------------------------------
# Heterogeneous rung of the ladder: manufactured solutions.  The analytic checks above are homogeneous, and
# adjointness / finite-difference gradients only show self-consistency; here a smooth u(x, t) = g(t) phi(x) is
# imposed on a smoothly VARYING c(x), the source term that makes it an exact solution of the continuous equations is
# evaluated in closed form and injected at every node, and the oracle's field must converge to u at the scheme's
# orders -- an independent solution for variable C, for the sponge and (further down) for the CPML's coefficients.
# ---------------------------------------------------------------------------
import _mms  # noqa: E402  (tests/_mms.py: the manufactured problems, 2-D and 3-D; shared with the GPU suite)

_MMS_X = _mms.X


def _mms_sponge_error(n, order, dt, power=8, sigma_max=150.0, prop=CPropagator, ndim=2):
    """Relative L2 error over all recorded samples of the sponge scheme against the manufactured solution
    u = sin^4(pi t / Tg) prod_d sin^p(pi x_d / X) of  u_tt + sigma(x) u_t = c(x)^2 (lap u + S); sigma is the quadratic
    ramp over a quarter of the box, where u is 1e-2 .. 1e-1 of its peak."""
    return _mms.error(_mms.sponge_case(n, dt, ndim, power, sigma_max), prop, order)


def test_manufactured_solution_heterogeneous_medium_with_sponge_second_order_in_time():
    """dt ~ h refinement, O(8): the error falls 4x per level (second order in time; the space error is far below) and
    ends below 1e-5 -- variable c and the damped update agree with an independent exact solution."""
    dt0 = 0.5 * fo.cfl_dt(2640.0, _MMS_X / 40, 2, 8)
    e = [_mms_sponge_error(n, 8, dt0 * 40.0 / (n - 1), power=6) for n in (41, 81, 161)]
    assert 3.6 < e[0] / e[1] < 4.4 and 3.6 < e[1] / e[2] < 4.4, e
    assert e[2] < 1e-5 and e[0] < 2e-4, e
    assert abs(_mms_sponge_error(41, 8, dt0, power=6, prop=fo.Propagator) - e[0]) < 1e-9 * e[0]  # NumPy form too


@pytest.mark.parametrize("order,ratio", [(2, 4.0), (4, 16.0)])
def test_manufactured_solution_space_orders(order, ratio):
    """Fixed tiny dt (time error ~1e-7): halving h divides the error by 2^order for O(2) and O(4) ..."""
    e = [_mms_sponge_error(n, order, 2.5e-5) for n in (21, 41, 81)]
    assert 0.85 * ratio < e[0] / e[1] < 1.15 * ratio and 0.85 * ratio < e[1] / e[2] < 1.15 * ratio, e


def test_manufactured_solution_space_order_eight():
    """... and by more than 2^7 for O(8) (191 measured: 3.3e-5 -> 1.7e-7, then the time error's floor)."""
    e = [_mms_sponge_error(n, 8, 2.5e-5) for n in (21, 41)]
    assert e[0] / e[1] > 128.0 and e[1] < 1e-6, e


def _mms_cpml_error(n, dt, alpha_max, order=8, sigma_max=150.0, power=8, prop=CPropagator, ndim=2):
    """The same manufactured u on the same medium with the CONVOLUTIONAL PML; the memory variables of the continuous PML
    system are obtained semi-analytically (tests/_mms.py: four linear ODEs per border node, 8th-order Runge-Kutta to
    1e-12), independently of the finite-difference scheme."""
    return _mms.error(_mms.cpml_case(n, dt, alpha_max, ndim, power, sigma_max), prop, order)


@pytest.mark.parametrize("alpha_max,last", [(60.0, 1e-4), (0.0, 3e-4)])
def test_manufactured_solution_heterogeneous_medium_with_cpml(alpha_max, last):
    """h -> h / 2 with dt -> dt / 4: the error falls 4x per level -- second order in h (the quadratic sigma ramp is
    only C^1 at the border's inner edge, which caps the space order there) plus first order in dt (the recursive
    convolution) -- and ends at 6.3e-5 (alpha_max = 60 / s; 1.8e-4 at alpha_max = 0): the CPML's a, b coefficients,
    the two memory variables and their coupling into q converge to the continuous PML system on a variable medium."""
    e = [_mms_cpml_error(n, dt, alpha_max) for n, dt in ((41, 1.6e-3), (81, 4e-4), (161, 1e-4))]
    assert 3.4 < e[0] / e[1] < 4.8 and 3.4 < e[1] / e[2] < 4.8, e
    assert e[2] < last, e
    e_np = _mms_cpml_error(41, 1.6e-3, alpha_max, prop=fo.Propagator)
    assert abs(e_np - e[0]) < 1e-9 * e[0]


def test_manufactured_solution_in_three_dimensions():
    """The same rung in 3-D (what the stream kernel computes): sponge, dt ~ h, 21^3 -> 41^3: 4x; CPML, dt ~ h^2: 4x."""
    dt0 = 0.5 * fo.cfl_dt(2740.0, _MMS_X / 20, 3, 8)
    e = [_mms_sponge_error(n, 8, dt0 * 20.0 / (n - 1), power=6, ndim=3) for n in (21, 41)]
    assert 3.5 < e[0] / e[1] < 4.5 and e[1] < 3e-4, e
    e = [_mms_cpml_error(n, dt, 60.0, ndim=3) for n, dt in ((21, 3.2e-3), (41, 8e-4))]
    assert 3.2 < e[0] / e[1] < 5.0 and e[1] < 3e-3, e


# ---------------------------------------------------------------------------
# second absorbing boundary: convolutional PML (abc="cpml")
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("shape,order", CASES)
@pytest.mark.parametrize("alpha", [0.0, 40.0])
def test_cpml_adjoint_identity(shape, order, alpha):
    """<F s, r> = <s, F^T r> to round-off: the adjoint sweep is the exact transpose of the CPML recursion,
    sources and receivers inside the border included."""
    c, h, dt, src, rec, w, r = _random_case(shape, order)
    p = fo.Propagator(c, h, dt, order, 5, abc="cpml", pml_alpha_max=alpha)
    d = p.forward(src, w, rec, save=False)
    a = p.adjoint(r, image=False)
    lhs, rhs = float(np.sum(d * r)), float(np.sum(w * a))
    assert abs(lhs - rhs) <= 1e-12 * max(abs(lhs), abs(rhs))


@pytest.mark.parametrize("shape,order", [((30, 26), 8), ((14, 12, 16), 4)])
def test_cpml_gradient_matches_finite_differences(shape, order):
    c, h, dt, src, rec, w, _ = _random_case(shape, order, nt=60, nsrc=1)
    w = fo.ricker(60, dt, 0.1 / dt / 4)[:, None]
    kw = dict(sigma_max=900.0, abc="cpml", pml_alpha_max=30.0)

    def data(cm):
        p = fo.Propagator(cm, h, dt, order, 5, **kw)
        return p, p.forward(src, w, rec)

    _, d_obs = data(c * 1.02)
    p, d = data(c)
    p.adjoint(d - d_obs)
    g = p.gradient()
    dc = np.random.default_rng(3).standard_normal(shape)
    eps = 1e-3
    Jp = 0.5 * np.sum((data(c + eps * dc)[1] - d_obs) ** 2)
    Jm = 0.5 * np.sum((data(c - eps * dc)[1] - d_obs) ** 2)
    fd = (Jp - Jm) / (2 * eps)
    assert abs(fd - np.sum(g * dc)) <= 1e-6 * abs(fd)


def test_cpml_reflects_far_less_than_the_sponge_at_equal_width():
    """Homogeneous 2-D medium, 10-cell border: seismograms near the border against a run on a grid so large that
    nothing comes back within the window.  The sponge leaves tens of per cent, the CPML a few 1e-4 (alpha = 0)."""
    h, order, c0, n, big, npml, f0 = 10.0, 8, 2000.0, 80, 260, 10, 12.0
    dt = 0.7 * fo.cfl_dt(c0, h, 2, order)
    nt = 380
    w = fo.ricker(nt, dt, f0)

    def run(nn, abc, width, alpha=0.0):
        off = (nn - n) // 2
        p = fo.Propagator(np.full((nn, nn), c0), h, dt, order, width, abc=abc, pml_alpha_max=alpha)
        rec = np.array([[off + npml + 2, off + x] for x in range(npml + 2, n - npml - 2, 4)])
        return p.forward(np.array([[off + n // 2, off + n // 2]]), w, rec, save=False)

    ref = run(big, "sponge", 0)
    e_sponge = rel(run(n, "sponge", npml), ref)
    e_cpml = rel(run(n, "cpml", npml), ref)
    e_cpml_a = rel(run(n, "cpml", npml, np.pi * f0), ref)
    assert e_sponge > 0.05 and e_cpml < 2e-3 and e_cpml_a < 1e-2 and e_cpml < e_sponge / 50


def test_cpml_is_stable_over_many_steps():
    c0, h, order = 2000.0, 10.0, 8
    dt = 0.7 * fo.cfl_dt(c0, h, 2, order)
    p = CPropagator(np.full((48, 48), c0), h, dt, order, 8, abc="cpml", pml_alpha_max=np.pi * 12.0)
    d = p.forward(np.array([[24, 24]]), fo.ricker(4000, dt, 12.0), np.array([[10, 24]]), save=False)
    assert np.isfinite(d).all() and np.abs(d[-300:]).max() < 1e-4 * np.abs(d).max()


@pytest.mark.parametrize("shape,order", [((30, 26), 8), ((30, 26), 2), ((14, 12, 16), 8), ((9, 40), 4)])
def test_c_oracle_cpml_matches_numpy_oracle(shape, order):
    """... including a grid thinner than two borders ((9, 40) with npml = 5: the borders overlap)."""
    c, h, dt, src, rec, w, r = _random_case(shape, order)
    kw = dict(abc="cpml", pml_alpha_max=25.0)
    p, q = fo.Propagator(c, h, dt, order, 5, **kw), CPropagator(c, h, dt, order, 5, **kw)
    assert rel(q.forward(src, w, rec), p.forward(src, w, rec)) < 1e-12
    assert rel(q.adjoint(r), p.adjoint(r)) < 1e-12
    assert rel(q.gradient(), p.gradient()) < 1e-12
