"""L-BFGS (two-loop recursion, Armijo backtracking + Wolfe-curvature expansion, box projection) for the model update.

The reference has no optimiser (its "optimiser" is exhaustive random search,
full_waveform_inversion.py:713); BASELINE.json configs[4] asks for 5 L-BFGS
iterations on the all-reduced gradient, which is what this drives.
"""
from __future__ import annotations

import json
import os

import numpy as np


def save_state(path, it, x, f, g, S, Y, log, evals, history=None, bounds=None):
    """Optimiser state after iteration ``it`` (SURVEY.md s.5 "checkpoint / resume": "optimiser state save per L-BFGS
    iteration"): model, misfit, gradient, the curvature pairs oldest first, the log, and the settings a continuation
    must share (``history``, ``bounds``).  Written to a temporary file and renamed, so an interrupted run never leaves
    a torn file behind.  In a multi-rank job only ONE rank should pass a path (every rank holds the same state)."""
    tmp = "%s.tmp.%d.npz" % (path, os.getpid())
    arrays = {"x": np.asarray(x), "g": np.asarray(g)}
    for i, (s, y) in enumerate(zip(S, Y)):
        arrays["s%d" % i] = np.asarray(s)
        arrays["y%d" % i] = np.asarray(y)
    meta = {"history": None if history is None else int(history),
            "bounds": None if bounds is None else [float(bounds[0]), float(bounds[1])]}
    np.savez(tmp, it=np.int64(it), f=np.float64(f), evals=np.int64(evals), npairs=np.int64(len(S)),
             log=np.frombuffer(json.dumps(log).encode(), np.uint8),
             meta=np.frombuffer(json.dumps(meta).encode(), np.uint8), **arrays)
    os.replace(tmp, path)


def load_state(path):
    """The dict :func:`save_state` wrote: ``it, f, evals, x, g, S, Y, log`` (+ ``history``, ``bounds`` when the file
    holds them)."""
    with np.load(path) as z:
        n = int(z["npairs"])
        st = {"it": int(z["it"]), "f": float(z["f"]), "evals": int(z["evals"]), "x": z["x"].copy(), "g": z["g"].copy(),
              "S": [z["s%d" % i].copy() for i in range(n)], "Y": [z["y%d" % i].copy() for i in range(n)],
              "log": json.loads(bytes(z["log"]).decode())}
        if "meta" in z.files:
            st.update(json.loads(bytes(z["meta"]).decode()))
        return st


def _checked_resume(resume, history, bounds, shape=None):
    """Load and validate a state for continuation: the bit-for-bit promise holds only if the continuation runs with
    the settings of the run that wrote the file, so differing ``history`` / ``bounds`` (when the file records them)
    and arrays of the wrong shape are errors, not silent changes; more pairs than ``history`` are trimmed (oldest
    first), which is what the writing run would have done."""
    st = load_state(resume) if isinstance(resume, (str, os.PathLike)) else dict(resume)
    if st.get("history") is not None and int(st["history"]) != int(history):
        raise ValueError("resume: the state was written with history=%d, this run has history=%d"
                         % (st["history"], int(history)))
    if "bounds" in st:
        b = None if bounds is None else [float(bounds[0]), float(bounds[1])]
        if st["bounds"] != b:
            raise ValueError("resume: the state was written with bounds=%r, this run has bounds=%r" % (st["bounds"], b))
    x = np.asarray(st["x"])
    if shape is not None and tuple(x.shape) != tuple(shape):
        raise ValueError("resume: the state holds a model of shape %r, the engine's grid is %r" % (x.shape, tuple(shape)))
    for name, a in [("g", st["g"])] + [("s%d" % i, s) for i, s in enumerate(st["S"])] + \
                   [("y%d" % i, y) for i, y in enumerate(st["Y"])]:
        if np.asarray(a).shape != x.shape:
            raise ValueError("resume: %s has shape %r, the model %r" % (name, np.asarray(a).shape, x.shape))
    if len(st["S"]) != len(st["Y"]):
        raise ValueError("resume: %d s-vectors but %d y-vectors" % (len(st["S"]), len(st["Y"])))
    st["S"], st["Y"] = list(st["S"])[-int(history):], list(st["Y"])[-int(history):]
    return st


def _search_failed(log):
    return bool(log) and log[-1].get("note") == "line search failed"


def lbfgs(fg, x0, maxiter=5, history=5, first_step=None, bounds=None, dot=None, c1=1e-4, max_ls=8,
          gtol=0.0, callback=None, c2=0.9, checkpoint=None, resume=None):
    """Minimise ``f`` given ``fg(x) -> (f, g)``.

    Line search: backtracking (safeguarded quadratic interpolation) until the Armijo condition holds; while no backtracking was needed and the slope
    along the direction is still steeper than ``c2`` times the initial one (the weak Wolfe curvature condition
    fails: the step is too short, typical of the first, hand-scaled iteration) the step is doubled instead, as
    long as the misfit keeps decreasing.  ``c2=None`` switches the expansion off (plain Armijo).

    ``first_step``: largest change of any component in the first trial step (the gradient of
    an FWI misfit has no natural scale).  ``dot(a, b)``: inner product (pass ``Engine.dot``
    for the GPU wave-shuffle reduction).  Returns ``(x, f, log)``.

    ``checkpoint``: path of a state file rewritten after every completed iteration (:func:`save_state`); in a
    multi-rank job pass it on ONE rank only (every rank holds the same state).
    ``resume``: such a path (or the dict of :func:`load_state`): the run continues after the iteration it holds --
    ``x0`` is ignored and no misfit is re-evaluated -- and, ``fg`` being deterministic, reproduces the
    uninterrupted run bit for bit (``maxiter`` counts iterations of the whole run, not of this call).
    """
    dot = dot or (lambda a, b: float(np.sum(np.multiply(a, b, dtype=np.float64))))  # no BLAS threads
    if int(history) < 1:
        raise ValueError("history must be >= 1")
    lo, hi = bounds if bounds is not None else (None, None)

    def project(x):
        return np.clip(x, lo, hi) if bounds is not None else x

    if resume is not None:
        st = _checked_resume(resume, history, bounds, None if x0 is None else np.shape(x0))
        x, f, g, S, Y = st["x"], st["f"], st["g"], list(st["S"]), list(st["Y"])
        log, evals, it0 = list(st["log"]), st["evals"], st["it"]
        if _search_failed(log):
            return x, f, log  # the run ended on a failed search: a continuation would only repeat it
    else:
        x = project(np.array(x0, copy=True))
        f, g = fg(x)
        _require_finite(f, float(np.abs(g).max()), dot(g, g), 0)
        S, Y, log = [], [], [{"iter": 0, "f": f, "evals": 1}]
        evals, it0 = 1, 0
        if checkpoint:
            save_state(checkpoint, 0, x, f, g, S, Y, log, evals, history, bounds)
    for it in range(it0 + 1, maxiter + 1):
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
        t, best, shrunk = 1.0, None, False
        for _ in range(max_ls):
            xn = project(x + t * p)
            fn, gn = fg(xn)
            evals += 1
            if fn <= f + c1 * t * gp and (best is None or fn < best[1]):
                best = (t, fn, xn, gn)
                if shrunk or c2 is None or not dot(gn, p) < c2 * gp:
                    break  # Armijo holds and (after backtracking, or by the slope) the step is long enough
                t *= 2.0   # still descending steeply: the step is too short
            elif best is not None:
                break      # the doubled step overshot: keep the last good one
            else:
                t = _backtrack(t, f, gp, fn)
                shrunk = True
        if best is None:
            log.append({"iter": it, "f": f, "evals": evals, "note": "line search failed"})
            if checkpoint:  # the terminal entry too: a resumed run must not repeat the failed search
                save_state(checkpoint, it - 1, x, f, g, S, Y, log, evals, history, bounds)
            break
        t, fn, xn, gn = best
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
        if checkpoint:
            save_state(checkpoint, it, x, f, g, S, Y, log, evals, history, bounds)
        if callback:
            callback(it, x, f, g)
    return x, f, log


def _backtrack(t, f0, gp, ft):
    """Next trial step after ``t`` failed the Armijo test: the minimiser of the parabola through ``f(0) = f0``,
    ``f'(0) = gp`` and ``f(t) = ft``, kept inside ``[t / 10, t / 2]`` (a misfit evaluation is a full sweep over
    the shots, so a badly scaled first step should cost two or three of them, not one per halving)."""
    curv = ft - f0 - gp * t
    tq = -gp * t * t / (2.0 * curv) if np.isfinite(curv) and curv > 0.0 else 0.5 * t
    return min(max(tq, 0.1 * t), 0.5 * t)


def _require_finite(f, gmax, gg, it):
    """A blown-up propagation must not pass for convergence: ``max |g|`` built on ``fmax`` ignores NaN
    (an all-NaN gradient reads 0 = "stationary"), so the squared norm is checked as well."""
    if not (np.isfinite(f) and np.isfinite(gmax) and np.isfinite(gg)):
        raise FloatingPointError("L-BFGS iteration %d: non-finite misfit or gradient (f=%r, max|g|=%r, g.g=%r); "
                                 "check the time step against the CFL limit of the current model" % (it, f, gmax, gg))


def lbfgs_device(engine, fg, x0, maxiter=5, history=5, first_step=None, bounds=None, c1=1e-4, max_ls=8,
                 gtol=0.0, callback=None, c2=0.9, checkpoint=None, resume=None):
    """The same iteration as :func:`lbfgs` with every model-sized vector resident on the GPU
    (``Engine.vec_*`` slots): per iteration only scalars cross PCIe.

    ``fg(x_slot, g_slot) -> f`` evaluates the misfit at the model in ``x_slot`` and leaves the
    gradient in ``g_slot`` (see ``shots.misfit_and_gradient_device``).  Returns ``(x, f, log)``
    with ``x`` downloaded once at the end.

    ``checkpoint`` / ``resume``: as in :func:`lbfgs` (the state file is the same: a run may be saved by one and
    resumed by the other); saving downloads the model, the gradient and the curvature pairs once per iteration
    (2 + 2 x history model-sized arrays over PCIe: ~0.1 s at 256^3 against ~7 s of shots per evaluation).
    """
    m = int(history)
    if m < 1:
        raise ValueError("history must be >= 1")
    X, G, XN, GN, P, XB, GB = 0, 1, 2, 3, 4, 5, 6  # XB / GB: the best trial point of the line search so far
    # m + 1 pair slots: the candidate pair is formed in a spare slot, so the oldest pair is evicted only
    # once the candidate has passed the curvature test
    S0, Y0 = 7, 7 + (m + 1)
    engine.vec_create(7 + 2 * (m + 1))
    pairs = []  # ring of (s_slot, y_slot), oldest first
    free = list(range(m + 1))

    def save(it, f, log, evals):
        save_state(checkpoint, it, engine.vec_download(X), f, engine.vec_download(G),
                   [engine.vec_download(s) for s, _ in pairs], [engine.vec_download(y) for _, y in pairs], log, evals,
                   m, bounds)

    if resume is not None:
        st = _checked_resume(resume, m, bounds, getattr(engine, "shape", None))
        if _search_failed(st["log"]):
            return np.asarray(st["x"]), st["f"], list(st["log"])
        engine.vec_upload(X, st["x"])
        engine.vec_upload(G, st["g"])
        for s_h, y_h in list(zip(st["S"], st["Y"]))[-m:]:
            k = free.pop(0)
            engine.vec_upload(S0 + k, s_h)
            engine.vec_upload(Y0 + k, y_h)
            pairs.append((S0 + k, Y0 + k))
        f, log, evals, it0 = st["f"], list(st["log"]), st["evals"], st["it"]
    else:
        engine.vec_upload(X, x0)
        if bounds is not None:
            engine.vec_clip(X, *bounds)
        f = fg(X, G)
        _require_finite(f, engine.vec_absmax(G), engine.vec_dot(G, G), 0)
        log = [{"iter": 0, "f": f, "evals": 1}]
        evals, it0 = 1, 0
        if checkpoint:
            save(0, f, log, evals)
    for it in range(it0 + 1, maxiter + 1):
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
        t, best, shrunk = 1.0, None, False
        for _ in range(max_ls):  # the same search as lbfgs(): Armijo backtracking, Wolfe-curvature expansion
            engine.vec_copy(XN, X)
            engine.vec_axpby(XN, -t, P, 1.0)
            if bounds is not None:
                engine.vec_clip(XN, *bounds)
            fn = fg(XN, GN)
            evals += 1
            if fn <= f + c1 * t * gp and (best is None or fn < best[1]):
                best = (t, fn)
                XB, XN = XN, XB
                GB, GN = GN, GB
                if shrunk or c2 is None or not -engine.vec_dot(GB, P) < c2 * gp:
                    break
                t *= 2.0
            elif best is not None:
                break
            else:
                t = _backtrack(t, f, gp, fn)
                shrunk = True
        if best is None:
            log.append({"iter": it, "f": f, "evals": evals, "note": "line search failed"})
            if checkpoint:
                save(it - 1, f, log, evals)
            break
        t, fn = best
        XN, XB = XB, XN
        GN, GB = GB, GN
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
        if checkpoint:
            save(it, f, log, evals)
        if callback:
            callback(it, X, f, G)
    return engine.vec_download(X), f, log
