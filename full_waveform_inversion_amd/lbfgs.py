"""L-BFGS (two-loop recursion, Armijo backtracking, box projection) for the model update.

The reference has no optimiser (its "optimiser" is exhaustive random search,
full_waveform_inversion.py:713); BASELINE.json configs[4] asks for 5 L-BFGS
iterations on the all-reduced gradient, which is what this drives.
"""
from __future__ import annotations

import numpy as np


def lbfgs(fg, x0, maxiter=5, history=5, first_step=None, bounds=None, dot=None, c1=1e-4, max_ls=8,
          gtol=0.0, callback=None):
    """Minimise ``f`` given ``fg(x) -> (f, g)``.

    ``first_step``: largest change of any component in the first trial step (the gradient of
    an FWI misfit has no natural scale).  ``dot(a, b)``: inner product (pass ``Engine.dot``
    for the GPU wave-shuffle reduction).  Returns ``(x, f, log)``.
    """
    dot = dot or (lambda a, b: float(np.sum(np.multiply(a, b, dtype=np.float64))))  # no BLAS threads
    if int(history) < 1:
        raise ValueError("history must be >= 1")
    lo, hi = bounds if bounds is not None else (None, None)

    def project(x):
        return np.clip(x, lo, hi) if bounds is not None else x

    x = project(np.array(x0, copy=True))
    f, g = fg(x)
    _require_finite(f, float(np.abs(g).max()), dot(g, g), 0)
    S, Y, log = [], [], [{"iter": 0, "f": f, "evals": 1}]
    evals = 1
    for it in range(1, maxiter + 1):
        if not float(np.abs(g).max()) > gtol:
            break  # stationary (or projected onto a bound everywhere)
        q = np.array(g, copy=True)
        al = []
        for s, y in zip(reversed(S), reversed(Y)):
            rho = 1.0 / dot(y, s)
            a = rho * dot(s, q)
            q -= a * y
            al.append((a, rho))
        if S:
            q *= dot(S[-1], Y[-1]) / dot(Y[-1], Y[-1])
        else:
            gmax = float(np.abs(g).max())
            q *= (first_step if first_step is not None else 1.0) / gmax
        for (s, y), (a, rho) in zip(zip(S, Y), reversed(al)):
            b = rho * dot(y, q)
            q += s * (a - b)
        p = -q
        gp = dot(g, p)
        if not gp < 0.0:  # not a descent direction: restart from steepest descent
            S, Y = [], []
            p = -g * ((first_step if first_step is not None else 1.0) / float(np.abs(g).max()))
            gp = dot(g, p)
        t = 1.0
        for _ in range(max_ls):
            xn = project(x + t * p)
            fn, gn = fg(xn)
            evals += 1
            if fn <= f + c1 * t * gp:
                break
            t *= 0.5
        else:
            log.append({"iter": it, "f": f, "evals": evals, "note": "line search failed"})
            break
        s, y = xn - x, gn - g
        if dot(s, y) > 1e-12 * np.sqrt(dot(s, s) * dot(y, y)):
            S.append(s)
            Y.append(y)
            if len(S) > history:
                S.pop(0)
                Y.pop(0)
        x, f, g = xn, fn, gn
        _require_finite(f, float(np.abs(g).max()), dot(g, g), it)
        log.append({"iter": it, "f": f, "evals": evals, "step": t})
        if callback:
            callback(it, x, f, g)
    return x, f, log


def _require_finite(f, gmax, gg, it):
    """A blown-up propagation must not pass for convergence: ``max |g|`` built on ``fmax`` ignores NaN
    (an all-NaN gradient reads 0 = "stationary"), so the squared norm is checked as well."""
    if not (np.isfinite(f) and np.isfinite(gmax) and np.isfinite(gg)):
        raise FloatingPointError("L-BFGS iteration %d: non-finite misfit or gradient (f=%r, max|g|=%r, g.g=%r); "
                                 "check the time step against the CFL limit of the current model" % (it, f, gmax, gg))


def lbfgs_device(engine, fg, x0, maxiter=5, history=5, first_step=None, bounds=None, c1=1e-4, max_ls=8,
                 gtol=0.0, callback=None):
    """The same iteration as :func:`lbfgs` with every model-sized vector resident on the GPU
    (``Engine.vec_*`` slots): per iteration only scalars cross PCIe.

    ``fg(x_slot, g_slot) -> f`` evaluates the misfit at the model in ``x_slot`` and leaves the
    gradient in ``g_slot`` (see ``shots.misfit_and_gradient_device``).  Returns ``(x, f, log)``
    with ``x`` downloaded once at the end.
    """
    m = int(history)
    if m < 1:
        raise ValueError("history must be >= 1")
    X, G, XN, GN, P = 0, 1, 2, 3, 4
    # m + 1 pair slots: the candidate pair is formed in a spare slot, so the oldest pair is evicted only
    # once the candidate has passed the curvature test
    S0, Y0 = 5, 5 + (m + 1)
    engine.vec_create(5 + 2 * (m + 1))
    engine.vec_upload(X, x0)
    if bounds is not None:
        engine.vec_clip(X, *bounds)
    f = fg(X, G)
    _require_finite(f, engine.vec_absmax(G), engine.vec_dot(G, G), 0)
    pairs = []  # ring of (s_slot, y_slot), oldest first
    free = list(range(m + 1))
    log = [{"iter": 0, "f": f, "evals": 1}]
    evals = 1
    for it in range(1, maxiter + 1):
        gmax = engine.vec_absmax(G)
        if not gmax > gtol:
            break
        engine.vec_copy(P, G)  # P holds q; the search direction is -P
        al = []
        for s, y in reversed(pairs):
            rho = 1.0 / engine.vec_dot(y, s)
            a = rho * engine.vec_dot(s, P)
            engine.vec_axpby(P, -a, y, 1.0)
            al.append((a, rho))
        if pairs:
            s, y = pairs[-1]
            engine.vec_axpby(P, 0.0, P, engine.vec_dot(s, y) / engine.vec_dot(y, y))
        else:
            engine.vec_axpby(P, 0.0, P, (first_step if first_step is not None else 1.0) / gmax)
        for (s, y), (a, rho) in zip(pairs, reversed(al)):
            b = rho * engine.vec_dot(y, P)
            engine.vec_axpby(P, a - b, s, 1.0)
        gp = -engine.vec_dot(G, P)
        if not gp < 0.0:  # not a descent direction: restart from steepest descent
            free += [s - S0 for s, _ in pairs]
            pairs = []
            engine.vec_copy(P, G)
            engine.vec_axpby(P, 0.0, P, (first_step if first_step is not None else 1.0) / gmax)
            gp = -engine.vec_dot(G, P)
        t = 1.0
        for _ in range(max_ls):
            engine.vec_copy(XN, X)
            engine.vec_axpby(XN, -t, P, 1.0)
            if bounds is not None:
                engine.vec_clip(XN, *bounds)
            fn = fg(XN, GN)
            evals += 1
            if fn <= f + c1 * t * gp:
                break
            t *= 0.5
        else:
            log.append({"iter": it, "f": f, "evals": evals, "note": "line search failed"})
            break
        k = free.pop(0)  # never empty: at most m of the m + 1 slots hold history
        s, y = S0 + k, Y0 + k
        engine.vec_copy(s, XN)
        engine.vec_axpby(s, -1.0, X, 1.0)
        engine.vec_copy(y, GN)
        engine.vec_axpby(y, -1.0, G, 1.0)
        sy = engine.vec_dot(s, y)
        if sy > 1e-12 * np.sqrt(engine.vec_dot(s, s) * engine.vec_dot(y, y)):
            pairs.append((s, y))
            if len(pairs) > m:  # the new pair is good: now the oldest one goes
                s_old, _ = pairs.pop(0)
                free.append(s_old - S0)
        else:
            free.append(k)
        X, XN = XN, X
        G, GN = GN, G
        f = fn
        _require_finite(f, engine.vec_absmax(G), engine.vec_dot(G, G), it)
        log.append({"iter": it, "f": f, "evals": evals, "step": t})
        if callback:
            callback(it, X, f, G)
    return engine.vec_download(X), f, log
