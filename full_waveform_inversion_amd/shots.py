"""Shot loop, shot-parallel sharding and the gradient exchange.

Shots are independent forward+adjoint problems on one model; only the gradient
(and the scalar misfit) is summed over them.  One process per GPU owns shots
``rank, rank + world, ...`` and the per-rank partial gradients are summed by a
single RCCL all-reduce on the device accumulators.  The
reference's only parallel pattern is the same shape -- independent Monte Carlo
samples per process, gather at the end (full_waveform_inversion.py:822-848).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class Shot:
    src_idx: np.ndarray          # (nsrc, ndim)
    wavelet: np.ndarray          # (nt,) or (nt, nsrc)
    rec_idx: np.ndarray          # (nrec, ndim)
    d_obs: np.ndarray | None = None  # (nt, nrec)
    # off-grid points (points.Spread): src_idx / rec_idx / wavelet then hold the expanded node lists and
    # the data are gathered back to the points; see Shot.at_coordinates
    src_spread: object = None
    rec_spread: object = None
    point_wavelet: np.ndarray | None = None  # (nt[, nsrc points]): the wavelets before spreading

    @classmethod
    def at_coordinates(cls, src_xyz, wavelet, rec_xyz, shape, d_obs=None):
        """A shot whose sources / receivers sit at fractional grid coordinates (multilinear interpolation)."""
        from .points import Spread
        S, R = Spread(src_xyz, shape), Spread(rec_xyz, shape)
        return cls(S.idx, S.scatter(wavelet), R.idx, d_obs, S, R, np.asarray(wavelet))

    def _on_device(self, engine):
        """off-grid points and an engine that spreads / gathers them on the device (fwi_forward_spread)"""
        return self.src_spread is not None and self.rec_spread is not None and hasattr(engine, "forward_at")

    def forward(self, engine, save):
        """Seismograms at this shot's receivers, ``(nt, nrec)``."""
        if self._on_device(engine):
            return engine.forward_at(None, (self.src_spread, self.point_wavelet), self.rec_spread, save=save)
        d = engine.forward(None, (self.src_idx, self.wavelet), self.rec_idx, save=save)
        return self.rec_spread.gather(d) if self.rec_spread is not None else d

    def adjoint(self, engine, residual):
        """Back-propagate a residual given at this shot's receivers (imaging into the engine's accumulator)."""
        if self._on_device(engine):
            engine.adjoint(None if residual is None else np.ascontiguousarray(residual))
            return
        r = self.rec_spread.scatter(residual) if self.rec_spread is not None else residual
        engine.adjoint(np.ascontiguousarray(r))


def inversion_engine(shape, h, dt, nt_max, **kw):
    """The :class:`Engine` an INVERSION should run on: ``update_form="increment"`` unless the caller says otherwise.

    In fp32 the standard update form meets north_star's 1e-5 on seismograms and on the gradient of a GIVEN residual,
    but end to end -- the engine forming its own residual d_syn - d_obs -- the forward error is amplified by |d| / |r|
    (configs[1] at full size: 6.3e-5; profiles/r03_parity.json).  The increment form carries the same recursion as
    (u, v = u - u_prev) and meets a flat 1e-5 end to end on every BASELINE config family
    (tests/test_gpu_parity.py::test_end_to_end_fp32_increment_form_flat_1e5) at +4 B/update in 3-D (157 vs 110 ms per
    256^3 shot-gradient) and +2.5 % in 2-D.  Forward-only modelling and the headline bench keep the standard form
    (``Engine``'s default).  fp64 engines need neither; options the increment form does not combine with (the bf16
    store, an explicit 2-D "stream" kernel) fall back to the standard form.
    """
    from .engine import Engine
    if "update_form" not in kw:
        fp32 = np.dtype(kw.get("dtype", "float32")) == np.dtype(np.float32)
        combinable = kw.get("store_dtype", "native") == "native" and not (len(shape) == 2 and kw.get("kernel") == "stream")
        # 2-D with the convolutional PML: the 4-steps-per-launch kernel that carries the border (step2d_fused_cpml) has
        # no increment form, and the standard form measures 6.5e-6 end to end there at full size (configs[1] + CPML)
        cpml2d = len(shape) == 2 and kw.get("abc", "sponge") == "cpml" and kw.get("npml", 0) > 0
        kw["update_form"] = "increment" if (fp32 and combinable and not cpml2d) else "standard"
    return Engine(shape, h, dt, nt_max, **kw)


def partition_shots(nshots, rank, world):
    """Round-robin: rank r of `world` owns shots r, r + world, ..."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    return list(range(rank, nshots, world))


class NoExchange:
    """Single process: nothing to sum."""
    rank, world = 0, 1

    def reduce(self, engine, misfit, wrt):
        return engine.gradient(wrt), misfit

    def reduce_device(self, engine, misfit):
        return misfit


class RcclExchange:
    """Production exchange: RCCL all-reduce (over xGMI) of the device-side accumulators.

    ``rdzv`` (:class:`rendezvous.Rendezvous`, or anything with ``rank`` / ``world`` / ``broadcast`` /
    ``allreduce``) is the control plane: it moves the 128-byte ncclUniqueId from rank 0 to everyone and
    agrees on whether every rank's communicator came up.  A communicator that fails on ANY rank is fatal
    on ALL of them (the ranks that succeeded abort theirs): there is no host-side fallback for the sum.
    """

    def __init__(self, engine, rdzv):
        self.rank, self.world = rdzv.rank, rdzv.world
        uid = rdzv.broadcast(type(engine).comm_unique_id() if self.rank == 0 else None)
        err = ""
        try:
            engine.comm_init(self.rank, self.world, uid)
            seen = engine.comm_info()
            if seen != (self.world, self.rank):
                err = "RCCL reports rank %d of %d, expected %d of %d" % (seen[1], seen[0], self.rank, self.world)
        except Exception as ex:  # FwiError: reported to every rank below, then raised
            err = str(ex)
        ok = rdzv.allreduce([0.0 if err else 1.0], "min")[0] == 1.0
        if not ok:
            try:
                engine.comm_abort()
            except Exception:
                pass
            raise RuntimeError("RCCL communicator did not come up on every rank"
                               + (" (this rank: %s)" % err if err else " (failed on another rank)"))
        self.rccl_ranks = seen[0]

    def reduce(self, engine, misfit, wrt):
        engine.allreduce_gradient()
        return engine.gradient(wrt), engine.allreduce_f64([misfit])[0]

    def reduce_device(self, engine, misfit):
        """Sum the device-side accumulators only; the gradient stays on the GPU."""
        engine.allreduce_gradient()
        return engine.allreduce_f64([misfit])[0]


class HostExchange:
    """Sum on host arrays over the control plane (:class:`rendezvous.Rendezvous`): the path the CPU-only
    multi-process tests exercise with the oracle-backed engine; same sharding and reduction semantics as
    :class:`RcclExchange`.  Not a fallback of it -- nothing in the product path constructs this."""

    def __init__(self, rdzv):
        self.rdzv = rdzv
        self.rank, self.world = rdzv.rank, rdzv.world

    def reduce(self, engine, misfit, wrt):
        g = self.rdzv.allreduce_array(engine.gradient(wrt))
        return g, self.rdzv.allreduce([misfit])[0]


class EnginePool:
    """Several engines (contexts + streams) on ONE GPU working through a rank's shots concurrently.

    A 2-D shot is latency-bound and cannot fill an MI355X; independent shots overlap when each has its own
    context and is driven by its own host thread (ctypes releases the GIL during the C-ABI calls).  Measured
    on configs[2] (32 shots of 1024^2 x 2000 steps, forward + adjoint + imaging): 0.44 s one after the other,
    0.35 s with two contexts, 0.35 - 0.38 s with three to six, 1.06 s with eight -- more streams than the
    runtime has hardware queues for serialise and pay for it; hence ``MAX_USEFUL``.  With the convolutional PML
    (``abc="cpml"``) a 1024^2 launch ends on its four corner tiles, and the overlap is what fills the idle CUs: 1.01 s
    one after the other, 0.65 s with two contexts, **0.55 s with three**, 1.24 s with four (round 3).  3-D shots are
    bandwidth-bound: use a pool of one.  Engines after the first only add into the first one's gradient
    accumulator, which also owns the RCCL communicator.
    """

    MAX_USEFUL = 6

    def __init__(self, make_engine, size):
        size = max(1, int(size))
        if size > self.MAX_USEFUL:
            import warnings
            warnings.warn("EnginePool of %d contexts on one GPU: beyond %d the streams share hardware queues and the "
                          "shots slow down (configs[2]: 0.35 s with 2 - 6 contexts, 1.06 s with 8)"
                          % (size, self.MAX_USEFUL), RuntimeWarning, stacklevel=2)
        self.engines = [make_engine() for _ in range(size)]

    @property
    def primary(self):
        return self.engines[0]

    def close(self):
        for e in self.engines:
            e.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def map_shots(self, indices, fn):
        """Run ``fn(engine, shot_index)`` for every index, engines working in parallel; returns the
        results in index order."""
        import threading
        indices = list(indices)
        out = [None] * len(indices)
        err = []

        def work(k):
            e = self.engines[k]
            try:
                for pos in range(k, len(indices), len(self.engines)):
                    out[pos] = fn(e, indices[pos])
            except BaseException as ex:  # re-raised in the caller's thread
                err.append(ex)

        threads = [threading.Thread(target=work, args=(k,)) for k in range(len(self.engines))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if err:
            raise err[0]
        return out


def _engines(engine):
    return engine.engines if isinstance(engine, EnginePool) else [engine]


def model_data(engine, model, shots, exchange=None):
    """Synthesise observed data for the shots this rank owns (in place, returns the shots)."""
    ex = exchange or NoExchange()
    for e in _engines(engine):
        e.set_model(model)

    def one(e, i):
        s = shots[i]
        s.d_obs = s.forward(e, save=False)

    mine = partition_shots(len(shots), ex.rank, ex.world)
    if isinstance(engine, EnginePool):
        engine.map_shots(mine, one)
    else:
        for i in mine:
            one(engine, i)
    return shots


def misfit_and_gradient(engine, model, shots, exchange=None, wrt="velocity", objective=None, device_l2=True):
    """J = sum_shots objective(F_s(model), d_obs,s) and dJ/dmodel, summed over all ranks.

    ``objective(d_syn, d_obs) -> (J, dJ/dd_syn)``; default least squares 1/2 ||d_syn - d_obs||^2
    (see objectives.py for the reference's similarity measures as misfits).

    ``device_l2`` (least squares only): form the residual and J on the device (``fwi_misfit_l2``) -- in the
    engine's dtype, i.e. with an fp32 engine ``d_obs`` is rounded to fp32 before the subtraction, which puts
    ~6e-8 |d| / |r| of relative noise on J and on the residual (visible to a line search only once |r| / |d|
    approaches 1e-6).  ``device_l2=False`` keeps the residual in fp64 on the host, ``d_obs`` exact.
    """
    from .objectives import l2
    objective = objective or l2
    ex = exchange or NoExchange()
    for e in _engines(engine):
        e.set_model(model)
        e.reset_gradient()
    misfit = _sweep_shots(engine, shots, ex, objective, device_l2)
    return ex.reduce(_engines(engine)[0], misfit, wrt)[::-1]


def _sweep_shots(engine, shots, ex, objective, device_l2=True):
    """forward + adjoint of this rank's shots; returns the misfit, gradients summed into the
    (primary) engine's accumulator."""
    from .objectives import l2

    def one(e, i):
        s = shots[i]
        if s.d_obs is None:
            raise ValueError("shot %d has no observed data on rank %d" % (i, ex.rank))
        d = s.forward(e, save=True)
        if device_l2 and objective is l2 and hasattr(e, "misfit_l2") and (s.rec_spread is None or s._on_device(e)):
            # least squares: residual and misfit are formed on the device (per node, or per off-grid point)
            j = e.misfit_l2(s.d_obs)
            e.adjoint(None)
            return j
        j, r = objective(d, s.d_obs)
        s.adjoint(e, r)
        return j

    mine = partition_shots(len(shots), ex.rank, ex.world)
    if isinstance(engine, EnginePool):
        misfit = float(sum(engine.map_shots(mine, one)))
        for other in engine.engines[1:]:
            engine.primary.gradient_add_from(other)
        return misfit
    return float(sum(one(engine, i) for i in mine))


def misfit_and_gradient_device(engine, model_slot, grad_slot, shots, exchange=None, wrt="velocity",
                               objective=None, device_l2=True):
    """Like :func:`misfit_and_gradient`, with the model read from and the gradient written to
    device-resident vectors (``Engine.vec_*``): no model-sized array crosses PCIe."""
    from .objectives import l2
    objective = objective or l2
    ex = exchange or NoExchange()
    engs = _engines(engine)
    engs[0].set_model_vec(model_slot)
    if len(engs) > 1:  # the optimiser's vectors live in the primary engine: hand the model over
        model = engs[0].vec_download(model_slot)
        for e in engs[1:]:
            e.set_model(model)
    for e in engs:
        e.reset_gradient()
    misfit = _sweep_shots(engine, shots, ex, objective, device_l2)
    misfit = ex.reduce_device(engs[0], misfit)
    engs[0].gradient_vec(grad_slot, wrt)
    return misfit
