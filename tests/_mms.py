"""Manufactured solutions on a heterogeneous medium (tests only): an imposed u(x, t) = g(t) phi(x) on a smoothly varying
c(x), the source term that makes it an EXACT solution of the continuous equations evaluated in closed form and injected at
every node, and the exact values at the receivers.  Independent of any finite-difference code: tests/test_oracle.py uses it
as the heterogeneous rung of the (unpinned) oracle's ladder, tests/test_gpu_round4.py to hold the GPU kernels against the
exact solution directly.  2-D and 3-D; the damped leapfrog (sponge) and the convolutional PML.

    sponge:  u_tt + sigma(x) u_t = c(x)^2 (lap u + S)                   (sigma = 2 d / dt: what A (2u - B u_prev + q) discretises)
    CPML:    psi_t = -lam psi - sigma u_x,  zeta_t = -lam zeta - sigma (u_xx + psi_x),  lam = sigma + alpha   (per axis)
             u_tt = c^2 (lap u + sum_d (d_d psi_d + zeta_d) + S)
For u = g(t) phi(x) the CPML's memory variables are convolutions of g with exp(-lam t): psi = -sigma phi_x G,
zeta = -sigma (phi_xx G - (sigma phi_x)_x H - sigma phi_x lam_x K) with G' = -lam G + g, H' = -lam H + G, Gl = dG/dlam
(Gl' = -lam Gl - G), K' = -lam K + Gl -- four scalar linear ODEs per distinct lam (one per border node of an axis),
integrated to 1e-12 by an 8th-order Runge-Kutta."""
import numpy as np

from oracle import fwi_oracle as fo

X, TG, TEND = 400.0, 0.2, 0.15   # box edge (m), period of g, final time (s)


def sink(t, k):
    """sin^k(t) and its first two derivatives with respect to t."""
    s, co = np.sin(t), np.cos(t)
    return s ** k, k * s ** (k - 1) * co, k * (k - 1) * s ** (k - 2) * co ** 2 - k * s ** k


def _outer(factors):
    """prod_d factors[d][i_d] as an ndim array (z first)."""
    out = factors[0]
    for f in factors[1:]:
        out = out[..., None] * f
    return out


def medium(n, ndim=2):
    x = np.arange(n) * (X / (n - 1))
    g = np.meshgrid(*([x] * ndim), indexing="ij")
    z, xx = g[0], g[-1]
    c = 2000.0 * (1 + 0.2 * np.sin(2 * np.pi * xx / X) * np.cos(2 * np.pi * z / X) + 0.1 * z / X)
    if ndim == 3:
        c = c + 2000.0 * 0.05 * np.sin(2 * np.pi * g[1] / X)
    return x, c


def points(n, ndim=2):
    src = np.stack(np.meshgrid(*([np.arange(n)] * ndim), indexing="ij"), -1).reshape(-1, ndim)
    st = max(1, (n - 1) // 20)   # receivers: the nodes of the coarsest level, at every level
    rec = np.stack(np.meshgrid(*([np.arange(0, n, st)] * ndim), indexing="ij"), -1).reshape(-1, ndim)
    return src, rec, st


def _common(n, dt, ndim, power, npml=None):
    h = X / (n - 1)
    npml = (n - 1) // 4 if npml is None else npml
    nt = int(round(TEND / dt))
    dt = TEND / nt
    x, c = medium(n, ndim)
    f0, f1, f2 = sink(np.pi * x / X, power)
    f1, f2 = f1 * np.pi / X, f2 * (np.pi / X) ** 2
    phi = _outer([f0] * ndim)
    lap = sum(_outer([f2 if a == d else f0 for a in range(ndim)]) for d in range(ndim))
    t = np.arange(nt) * dt
    src, rec, st = points(n, ndim)
    sub = tuple(slice(None, None, st) for _ in range(ndim))
    exact = sink(np.pi * (t + dt) / TG, 4)[0][:, None] * phi[sub].ravel()[None]
    return dict(h=h, npml=npml, nt=nt, dt=dt, x=x, c=c, f=(f0, f1, f2), phi=phi, lap=lap, t=t, src=src, rec=rec, exact=exact)


def sponge_case(n, dt, ndim=2, power=8, sigma_max=150.0, npml=None):
    """Inputs and exact receiver values of the sponge problem: dict(c, h, dt, nt, npml, sigma_max, src, wav, rec, exact).
    `npml`: border width in cells (default: a quarter of the box)."""
    k = _common(n, dt, ndim, power, npml)
    prof = fo.damping_profiles((n,) * ndim, k["npml"], sigma_max, k["dt"])
    sig = sum(p.reshape([-1 if a == d else 1 for a in range(ndim)]) for d, p in enumerate(prof)) * 2.0 / k["dt"]
    g, g1, g2 = sink(np.pi * k["t"] / TG, 4)
    g1, g2 = g1 * np.pi / TG, g2 * (np.pi / TG) ** 2
    S = (g2[:, None] * k["phi"].ravel() + g1[:, None] * (sig * k["phi"]).ravel()) / k["c"].ravel() ** 2 \
        - g[:, None] * k["lap"].ravel()
    return dict(c=k["c"], h=k["h"], dt=k["dt"], nt=k["nt"], npml=k["npml"], sigma_max=sigma_max, src=k["src"],
                wav=S * k["h"] ** ndim, rec=k["rec"], exact=k["exact"], kw={})


def cpml_case(n, dt, alpha_max, ndim=2, power=8, sigma_max=150.0, npml=None):
    from scipy.integrate import solve_ivp
    k = _common(n, dt, ndim, power, npml)
    x, nt, t = k["x"], k["nt"], k["t"]
    L = k["npml"] * k["h"]
    xi = np.maximum(0.0, np.maximum(L - x, x - (X - L))) / L           # dist / npml as a function of the coordinate
    sgn = np.where(x < L, -1.0, np.where(x > X - L, 1.0, 0.0))          # L d(xi)/dx
    sig, dsig = sigma_max * xi ** 2, 2.0 * sigma_max * xi * sgn / L
    lam, dlam = sig + alpha_max * (1.0 - xi), dsig - alpha_max * sgn / L
    ker, cache = np.zeros((4, nt, n)), {}
    for i in np.nonzero(xi > 0)[0]:
        key = round(float(lam[i]), 9)
        if key not in cache:
            li = float(lam[i])
            cache[key] = solve_ivp(lambda tt, y: [-li * y[0] + np.sin(np.pi * tt / TG) ** 4, -li * y[1] + y[0],
                                                  -li * y[2] - y[0], -li * y[3] + y[2]],
                                   (0.0, TEND), [0.0] * 4, method="DOP853", rtol=1e-12, atol=1e-16, t_eval=t).y
        ker[:, :, i] = cache[key]
    G, H, Gl, K = ker
    f0, f1, f2 = k["f"]
    sf1, dsf1 = sig * f1, dsig * f1 + sig * f2
    # one axis' d psi / dx + zeta as a function of (t, its coordinate), for a unit transverse factor
    term = (-(dsf1[None] * G) - (sf1 * dlam)[None] * Gl) - sig[None] * (f2[None] * G - dsf1[None] * H - (sf1 * dlam)[None] * K)
    g, _, g2 = sink(np.pi * t / TG, 4)
    g2 = g2 * (np.pi / TG) ** 2
    shape_t = (nt,) + (1,) * ndim
    S = g2.reshape(shape_t) * k["phi"][None] / k["c"][None] ** 2 - g.reshape(shape_t) * k["lap"][None]
    for d in range(ndim):  # the axis' term times the other axes' f
        fac = term.reshape((nt,) + tuple(n if a == d else 1 for a in range(ndim)))
        for a in range(ndim):
            if a != d:
                fac = fac * f0.reshape((1,) + tuple(n if b == a else 1 for b in range(ndim)))
        S = S - fac
    return dict(c=k["c"], h=k["h"], dt=k["dt"], nt=nt, npml=k["npml"], sigma_max=sigma_max, src=k["src"],
                wav=S.reshape(nt, -1) * k["h"] ** ndim, rec=k["rec"], exact=k["exact"],
                kw=dict(abc="cpml", pml_alpha_max=alpha_max))


def error(case, prop, order=8):
    """Relative L2 error over all recorded samples of `prop` (an oracle class, or anything with its constructor and
    forward signature) against the exact solution."""
    p = prop(case["c"], case["h"], case["dt"], order, case["npml"], sigma_max=case["sigma_max"], **case["kw"])
    d = np.asarray(p.forward(case["src"], case["wav"], case["rec"], save=False), np.float64)
    return float(np.linalg.norm(d - case["exact"]) / np.linalg.norm(case["exact"]))
