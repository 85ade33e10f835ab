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

    @classmethod
    def at_coordinates(cls, src_xyz, wavelet, rec_xyz, shape, d_obs=None):
        """A shot whose sources / receivers sit at fractional grid coordinates (multilinear interpolation)."""
        from .points import Spread
        S, R = Spread(src_xyz, shape), Spread(rec_xyz, shape)
        return cls(S.idx, S.scatter(wavelet), R.idx, d_obs, S, R)

    def forward(self, engine, save):
        """Seismograms at this shot's receivers, ``(nt, nrec)``."""
        d = engine.forward(None, (self.src_idx, self.wavelet), self.rec_idx, save=save)
        return self.rec_spread.gather(d) if self.rec_spread is not None else d

    def adjoint(self, engine, residual):
        """Back-propagate a residual given at this shot's receivers (imaging into the engine's accumulator)."""
        r = self.rec_spread.scatter(residual) if self.rec_spread is not None else residual
        engine.adjoint(np.ascontiguousarray(r))


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

    ``broadcast`` moves the 128-byte ncclUniqueId from rank 0 to everyone over whatever
    control plane the launcher offers (bench.py uses torch.distributed/gloo for it).
    """

    def __init__(self, engine, rank, world, broadcast):
        self.rank, self.world = rank, world
        uid = type(engine).comm_unique_id() if rank == 0 else None
        engine.comm_init(rank, world, broadcast(uid))

    def reduce(self, engine, misfit, wrt):
        engine.allreduce_gradient()
        return engine.gradient(wrt), engine.allreduce_f64([misfit])[0]

    def reduce_device(self, engine, misfit):
        """Sum the device-side accumulators only; the gradient stays on the GPU."""
        engine.allreduce_gradient()
        return engine.allreduce_f64([misfit])[0]


class HostExchange:
    """Sum on host arrays through a torch.distributed process group (gloo): the path the
    CPU-only multi-process tests exercise; same sharding and reduction semantics."""

    def __init__(self, dist):
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def reduce(self, engine, misfit, wrt):
        import torch
        g = np.ascontiguousarray(engine.gradient(wrt))
        tg = torch.from_numpy(g)
        self.dist.all_reduce(tg)
        tj = torch.tensor([misfit], dtype=torch.float64)
        self.dist.all_reduce(tj)
        return g, float(tj[0])


class EnginePool:
    """Several engines (contexts + streams) on ONE GPU working through a rank's shots concurrently.

    A 2-D shot is latency-bound and cannot fill an MI355X (1024^2: 7 us/step alone, 3.5 us/step/shot
    with four in flight); independent shots overlap when each has its own context and is driven
    by its own host thread (ctypes releases the GIL during the C-ABI calls).  3-D shots are
    bandwidth-bound: use a pool of one.  Engines after the first only add into the first one's
    gradient accumulator, which also owns the RCCL communicator.
    """

    def __init__(self, make_engine, size):
        self.engines = [make_engine() for _ in range(max(1, int(size)))]

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


def misfit_and_gradient(engine, model, shots, exchange=None, wrt="velocity", objective=None):
    """J = sum_shots objective(F_s(model), d_obs,s) and dJ/dmodel, summed over all ranks.

    ``objective(d_syn, d_obs) -> (J, dJ/dd_syn)``; default least squares 1/2 ||d_syn - d_obs||^2
    (see objectives.py for the reference's similarity measures as misfits).
    """
    from .objectives import l2
    objective = objective or l2
    ex = exchange or NoExchange()
    for e in _engines(engine):
        e.set_model(model)
        e.reset_gradient()
    misfit = _sweep_shots(engine, shots, ex, objective)
    return ex.reduce(_engines(engine)[0], misfit, wrt)[::-1]


def _sweep_shots(engine, shots, ex, objective):
    """forward + adjoint of this rank's shots; returns the misfit, gradients summed into the
    (primary) engine's accumulator."""
    def one(e, i):
        s = shots[i]
        if s.d_obs is None:
            raise ValueError("shot %d has no observed data on rank %d" % (i, ex.rank))
        d = s.forward(e, save=True)
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
                               objective=None):
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
    misfit = _sweep_shots(engine, shots, ex, objective)
    misfit = ex.reduce_device(engs[0], misfit)
    engs[0].gradient_vec(grad_slot, wrt)
    return misfit
