#!/usr/bin/env python3
"""Run one of BASELINE.json's inversion configs end to end (shots -> gradient -> L-BFGS).

Single GPU:   python tools/run_config.py --config cfg3 --scale 0.25
Several GPUs: python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 \
                  --master-port P tools/run_config.py --config cfg5 --scale 0.5 --iters 5
Shots are sharded rank::world, the gradient is summed by one RCCL all-reduce per evaluation.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from full_waveform_inversion_amd import Engine, shots as sh, workloads  # noqa: E402
from full_waveform_inversion_amd.lbfgs import lbfgs, lbfgs_device  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg3", choices=["cfg3", "cfg5"])
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--shots", type=int, default=0, help="0 = the config's own count (32 / 64)")
    ap.add_argument("--iters", type=int, default=0, help="L-BFGS iterations; 0 = one gradient only")
    ap.add_argument("--host-lbfgs", action="store_true", help="keep the optimiser vectors on the host")
    ap.add_argument("--pool", type=int, default=0,
                    help="engines sharing this GPU's shots concurrently; 0 = auto (2 in 2-D, 1 in 3-D)")
    ap.add_argument("--abc", default="sponge", choices=["sponge", "cpml"], help="absorbing border (fwi_config.abc)")
    ap.add_argument("--image-stride", type=int, default=1,
                    help="imaging condition every S-th step (fwi_config.image_stride)")
    ap.add_argument("--update-form", default="auto", choices=["auto", "increment", "standard"],
                    help="fp32 update form; auto = shots.inversion_engine's choice (increment: 1e-5 end to end)")
    ap.add_argument("--launch-mode", default="auto", choices=["auto", "stream", "graph"])
    ap.add_argument("--checkpoint", default="", help="optimiser state file, rewritten after every iteration (rank 0 only)")
    ap.add_argument("--resume", default="", help="continue from such a state file (every rank reads it)")
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    from full_waveform_inversion_amd import _lib as _fl
    ndev = _fl.device_count()  # a launcher may have narrowed this process to one visible GPU
    if ndev > 0:
        local %= ndev
    kw = {"nshots": a.shots} if a.shots else {}
    w = workloads.CONFIGS[a.config](a.scale, **kw)
    wav = w.wavelet()
    shots = [sh.Shot(w.src_idx[i:i + 1], wav, w.rec_idx) for i in range(len(w.src_idx))]
    psize = a.pool or ((3 if a.abc == "cpml" else 2) if w.ndim == 2 else 1)  # (measured: shots.EnginePool)
    from full_waveform_inversion_amd import default_sigma_max
    sigma = default_sigma_max(float(w.c.max()), w.h, w.npml)
    uf = {} if a.update_form == "auto" else {"update_form": a.update_form}
    pool = sh.EnginePool(lambda: sh.inversion_engine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml, device=local,
                                                     sigma_max=sigma, image_stride=a.image_stride, abc=a.abc,
                                                     launch_mode=a.launch_mode, **uf,
                                                     pml_alpha_max=(3.14159 * w.f0 if a.abc == "cpml" else 0.0)), psize)
    e = pool.primary
    ex = sh.NoExchange()
    rdzv = None
    # FWI_RUN_FORCE_EXCHANGE=1: take the N > 1 path (rendezvous, RCCL communicator, all-reduce per evaluation) with one
    # rank too -- how a one-GPU box rehearses what the driver's launch line does for N = 2, 4, 8
    if world > 1 or os.environ.get("FWI_RUN_FORCE_EXCHANGE") == "1":
        # control plane: the package's stdlib rendezvous on MASTER_ADDR / MASTER_PORT (no torch in this process)
        from full_waveform_inversion_amd.rendezvous import Rendezvous
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        rdzv = Rendezvous.from_env()
        e.set_model(w.c.astype(np.float32))
        ex = sh.RcclExchange(e, rdzv)  # raises on every rank if any rank's communicator fails
    sh.model_data(pool, w.c.astype(np.float32), shots, ex)
    t0 = time.perf_counter()
    evals = [0]

    def fg(m):
        evals[0] += 1
        return sh.misfit_and_gradient(pool, m, shots, ex)

    m0 = w.c_init.astype(np.float32)
    ckpt = a.checkpoint if (a.checkpoint and rank == 0) else None  # every rank holds the same state: one writer
    bounds = (0.5 * float(w.c.min()), 1.5 * float(w.c.max()))
    if a.iters > 0 and not a.host_lbfgs:
        def fg_dev(xs, gs):
            evals[0] += 1
            return sh.misfit_and_gradient_device(pool, xs, gs, shots, ex)
        _, _, log = lbfgs_device(e, fg_dev, m0, maxiter=a.iters, history=5, first_step=0.02 * float(m0.max()),
                                 bounds=bounds, checkpoint=ckpt, resume=a.resume or None)
    elif a.iters > 0:
        _, _, log = lbfgs(fg, m0, maxiter=a.iters, history=5, first_step=0.02 * float(m0.max()),
                          bounds=bounds, dot=e.dot, checkpoint=ckpt, resume=a.resume or None)
    else:
        J, g = fg(m0)
        log = [{"iter": 0, "f": J, "gnorm": float(np.sqrt(e.dot(g, g)))}]
    el = time.perf_counter() - t0
    if rank == 0:
        upd = 2 * evals[0] * len(shots) * w.updates_per_shot  # forward + adjoint sweeps
        print(json.dumps({"config": w.name, "shape": list(w.shape), "nt": w.nt, "shots": len(shots),
                          "n_gpus": world, "rccl_ranks": getattr(ex, "rccl_ranks", None), "engines_per_gpu": psize, "evaluations": evals[0], "seconds": round(el, 3),
                          "Gpts_per_s_fwd_plus_adj": round(upd / el / 1e9, 2), "kernel": e.kernel_name,
                          "update_form": e.update_form, "abc": a.abc, "launch_mode": a.launch_mode,
                          "log": log}))
    pool.close()
    if rdzv is not None:
        rdzv.barrier()
        rdzv.close()


if __name__ == "__main__":
    main()
