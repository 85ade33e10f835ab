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
    lo, hi = bounds if bounds is not None else (None, None)

    def project(x):
        return np.clip(x, lo, hi) if bounds is not None else x

    x = project(np.array(x0, copy=True))
    f, g = fg(x)
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
        log.append({"iter": it, "f": f, "evals": evals, "step": t})
        if callback:
            callback(it, x, f, g)
    return x, f, log
