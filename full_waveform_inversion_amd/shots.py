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


def model_data(engine, model, shots, exchange=None):
    """Synthesise observed data for the shots this rank owns (in place, returns the shots)."""
    ex = exchange or NoExchange()
    engine.set_model(model)
    for i in partition_shots(len(shots), ex.rank, ex.world):
        s = shots[i]
        s.d_obs = engine.forward(None, (s.src_idx, s.wavelet), s.rec_idx, save=False)
    return shots


def misfit_and_gradient(engine, model, shots, exchange=None, wrt="velocity", objective=None):
    """J = sum_shots objective(F_s(model), d_obs,s) and dJ/dmodel, summed over all ranks.

    ``objective(d_syn, d_obs) -> (J, dJ/dd_syn)``; default least squares 1/2 ||d_syn - d_obs||^2
    (see objectives.py for the reference's similarity measures as misfits).
    """
    from .objectives import l2
    objective = objective or l2
    ex = exchange or NoExchange()
    engine.set_model(model)
    engine.reset_gradient()
    misfit = 0.0
    for i in partition_shots(len(shots), ex.rank, ex.world):
        s = shots[i]
        if s.d_obs is None:
            raise ValueError("shot %d has no observed data on rank %d" % (i, ex.rank))
        d = engine.forward(None, (s.src_idx, s.wavelet), s.rec_idx, save=True)
        j, r = objective(d, s.d_obs)
        misfit += j
        engine.adjoint(r)
    return ex.reduce(engine, misfit, wrt)[::-1]


def misfit_and_gradient_device(engine, model_slot, grad_slot, shots, exchange=None, wrt="velocity",
                               objective=None):
    """Like :func:`misfit_and_gradient`, with the model read from and the gradient written to
    device-resident vectors (``Engine.vec_*``): no model-sized array crosses PCIe."""
    from .objectives import l2
    objective = objective or l2
    ex = exchange or NoExchange()
    engine.set_model_vec(model_slot)
    engine.reset_gradient()
    misfit = 0.0
    for i in partition_shots(len(shots), ex.rank, ex.world):
        s = shots[i]
        if s.d_obs is None:
            raise ValueError("shot %d has no observed data on rank %d" % (i, ex.rank))
        d = engine.forward(None, (s.src_idx, s.wavelet), s.rec_idx, save=True)
        j, r = objective(d, s.d_obs)
        misfit += j
        engine.adjoint(r)
    misfit = ex.reduce_device(engine, misfit)
    engine.gradient_vec(grad_slot, wrt)
    return misfit
