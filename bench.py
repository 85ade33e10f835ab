#!/usr/bin/env python3
"""Headline benchmark: grid-point updates/s of the 3-D 256^3 O(8) acoustic stencil.

One "step" = one full shot of BASELINE.json configs[3] (256^3 constant velocity,
1000 time steps, O(8), fp32) through fwi_forward() on one MI355X.  With N > 1
(launched by torch.distributed.run, one process per GPU) every rank runs its
own shot per step -- shots are independent -- and the K steps of the timed
region form one gradient evaluation over N*K shots, closed by the path's one
real exchange: ONE RCCL all-reduce of the model-sized gradient accumulator
("an RCCL all-reduce ... only for the final gradient sum", north_star), inside
the timed region.  The reported value is the aggregate over ranks.  torch is used only for the
rendezvous / barrier / max-over-ranks (gloo); the data path is HIP + RCCL
behind the C-ABI.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_UPDATE = 16        # read u_cur, u_prev, C; write u_next (fp32)


def cpu_baseline(w, sample_steps):
    """The oracle's OpenMP C port (fp64) on the same grid for the first `sample_steps` steps."""
    from oracle.c_oracle import CPropagator
    from oracle.c_oracle import default_threads
    cores = int(os.environ.get("OMP_NUM_THREADS", default_threads()))
    os.environ["OMP_NUM_THREADS"] = str(cores)
    p = CPropagator(w.c, w.h, w.dt, w.order, w.npml)
    sample_steps = min(sample_steps, w.nt)
    wav = w.wavelet(np.float64)[:sample_steps]
    p.forward(w.src_idx, wav[:2], w.rec_idx, save=False)  # warm-up / page-in
    t0 = time.perf_counter()
    p.forward(w.src_idx, wav, w.rec_idx, save=False)
    el = time.perf_counter() - t0
    out = {"value": int(np.prod(w.shape)) * sample_steps / el / 1e9, "unit": "Gpts/s", "cores": cores,
           "kind": "port",
           "sample": "same %s grid, first %d of %d time steps, fp64 OpenMP C port of the build's oracle "
                     "(the reference has no such path), %.1f s" % ("x".join(map(str, w.shape)),
                                                                   sample_steps, w.nt, el)}
    # the NumPy form of the same oracle (what a "NumPy CPU path" of this scheme is): sliced-array stencil,
    # effectively one core; two steps of the same grid are enough for a rate
    from oracle import fwi_oracle as fo
    q = fo.Propagator(w.c, w.h, w.dt, w.order, w.npml)
    nsteps = 8
    t0 = time.perf_counter()
    q.forward(w.src_idx, wav[:nsteps], w.rec_idx, save=False)
    el_np = time.perf_counter() - t0
    out["numpy_1core"] = {"value": int(np.prod(w.shape)) * nsteps / el_np / 1e9, "unit": "Gpts/s", "cores": 1,
                          "sample": "%d time steps of the NumPy oracle on the same grid, %.1f s" % (nsteps, el_np)}
    return out


def measured_traffic(kernel_name, shape):
    """HBM-side bytes per launch from the committed rocprofv3 PMC passes (profiles/), or None.

    PMC counters cannot be collected inside this process; the numbers come from separate
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this same command, condensed by
    tools/summarize_profile.py.  Only quoted when kernel and grid match what is being run.
    """
    path = os.path.join(ROOT, "profiles", "r01_step3d_stream_%d.json" % shape[0])
    try:
        prof = json.load(open(path))
    except (OSError, ValueError):
        return None, None
    if kernel_name not in prof.get("kernel", "") or len(set(shape)) != 1:
        return None, None
    return prof["traffic_bytes_per_launch"], os.path.relpath(path, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--grid", type=int, default=256, help="cube edge (256 = BASELINE config)")
    ap.add_argument("--nt", type=int, default=1000)
    ap.add_argument("--npml", type=int, default=0)
    ap.add_argument("--kernel", default="auto")
    ap.add_argument("--zchunk", type=int, default=0)
    ap.add_argument("--image-stride", type=int, default=1,
                    help="gradient mode: store / correlate the forward term every S-th step (fwi_config.image_stride)")
    ap.add_argument("--mode", default="forward", choices=["forward", "gradient"],
                    help="forward: the headline stencil run; gradient: forward(save) + adjoint(imaging) per step")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal on a 1-GPU box: every rank uses device 0 (RCCL refuses duplicate GPUs, "
                         "so this exercises the flagged fallback exchange)")
    ap.add_argument("--cpu-steps", type=int, default=1000,
                    help="time steps of the CPU baseline sample (1000 = the whole shot, ~12 s on 16 cores); 0 = skip")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if args.single_device else int(os.environ.get("LOCAL_RANK", "0"))
    if not args.single_device:
        # a launcher may already have narrowed this process to its own GPU (HIP_VISIBLE_DEVICES /
        # ROCR_VISIBLE_DEVICES): then the only visible ordinal is 0
        from full_waveform_inversion_amd import _lib as _fl
        ndev = _fl.device_count()
        if ndev > 0:
            local %= ndev
    if world != args.gpus:
        if rank == 0 and world > 1:
            print("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run "
                     "--nproc-per-node %d" % (args.gpus, args.gpus))
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)

    from full_waveform_inversion_amd import Engine, workloads

    w = workloads.cfg4(args.grid / 256.0, npml=args.npml)
    w.nt = args.nt
    wav = w.wavelet()
    model = w.c.astype(np.float32)
    e = Engine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml, device=local, kernel=args.kernel,
               zchunk=args.zchunk, image_stride=args.image_stride)
    t_sm = time.perf_counter()
    e.set_model(model)
    set_model_ms = 1e3 * (time.perf_counter() - t_sm)  # host padding + H2D, once per model (not timed)
    exchange = "none"
    if world > 1:
        # The gradient sum goes over RCCL (xGMI), straight from the C-ABI.  Should the communicator
        # fail to come up on every rank, the sum is still performed -- over the gloo control plane,
        # on host copies -- and the JSON line says so; nothing is skipped silently.
        from full_waveform_inversion_amd import FwiError
        try:
            ids = [Engine.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            # ncclCommInitRank in a helper thread (ctypes drops the GIL): a communicator that neither
            # comes up nor fails within the limit is treated like one that failed
            import threading
            res = {}

            def init():
                try:
                    e.comm_init(rank, world, ids[0])
                    res["ok"] = True
                except FwiError as ex:
                    res["err"] = str(ex)

            th = threading.Thread(target=init, daemon=True)
            th.start()
            th.join(float(os.environ.get("FWI_COMM_INIT_TIMEOUT", "180")))
            if res.get("ok"):
                ok, why = 1, ""
            else:
                ok, why = 0, res.get("err", "ncclCommInitRank did not return within the time limit")
        except FwiError as ex:
            ok, why = 0, str(ex)
        import torch
        flag = torch.tensor([ok])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag[0]) == 1:
            exchange = "one rccl allreduce of the gradient accumulator after the K shots (timed)"
        else:
            exchange = "GLOO FALLBACK (rccl communicator failed: %s): host allreduce after the K shots (timed)" % (
                why or "on another rank")
            if rank == 0:
                print("warning: " + exchange, file=sys.stderr)

    def exchange_gradient():
        if exchange.startswith("one rccl"):
            e.allreduce_gradient()
        elif world > 1:
            import torch
            t = torch.from_numpy(e.gradient("slowness2"))
            dist.all_reduce(t)

    grad = args.mode == "gradient"

    def step():
        d = e.forward(None, (w.src_idx, wav), w.rec_idx, save=grad)
        ms = e.last_loop_ms()
        if grad:
            e.adjoint(d)  # residual = the data themselves: same work as any residual
            ms += e.last_loop_ms()
        return ms

    for _ in range(args.warmup):
        step()
    if world > 1:
        exchange_gradient()  # warm the communicator (first collective sets up the rings)
    if dist is not None:
        dist.barrier()
    e.synchronize()
    t0 = time.perf_counter()
    loop_ms = [step() for _ in range(args.steps)]
    if world > 1:
        exchange_gradient()  # the shot loop's one exchange: sum of the per-rank gradients
    e.synchronize()
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t[0])

    if rank == 0:
        npts = int(np.prod(w.shape))
        sweeps = 2 if grad else 1            # forward + adjoint sweep per shot
        # forward+save 20 B; adjoint + paired imaging 24 B (SURVEY s.8d prices the unpaired form at 28 B)
        S = max(1, args.image_stride)
        if grad and S > 1:  # every S-th step stores q (+4 B) / reads q and read-modify-writes g (+12 B)
            bpu = ((16 + 4.0 / S) + (16 + 12.0 / S)) / 2.0
        else:
            bpu = (20 + 24) / 2.0 if grad else BYTES_PER_UPDATE
        updates = npts * w.nt * sweeps
        value = world * args.steps * updates / el / 1e9
        kern_us = 1e3 * float(np.mean(loop_ms)) / (w.nt * sweeps)  # avg launch-to-launch time per step kernel
        achieved = bpu * npts / (kern_us * 1e-6) / 1e9
        traffic, traffic_src = (None, None) if grad else measured_traffic(e.kernel_name, w.shape)
        out = {
            "metric": "stencil grid-point-updates/sec (Gpts/s), 3-D O(8) acoustic" +
                      (" (forward+adjoint sweeps)" if grad else ""),
            "value": round(value, 3), "unit": "Gpts/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * el / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "set_model_ms": round(set_model_ms, 1),
            "config": {"workload": "configs[3]: 3-D %s constant velocity, 1 shot/GPU/step, %d time steps, "
                                   "O(8), npml=%d" % ("x".join(map(str, w.shape)), w.nt, w.npml),
                       "kernel": e.kernel_name, "image_stride": S, "parallelism": "shot-parallel x%d" % world,
                       "exchange": exchange},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": bpu * npts,
                         "kernel_avg_us": round(kern_us, 2),
                         "note": "algorithmic %g B/update x %d updates per launch / HIP-event time of the "
                                 "%d-launch loop(s); wavefield working set 3 x %d MiB" % (bpu, npts, w.nt * sweeps,
                                                                                         npts * 4 >> 20)},
        }
        if args.cpu_steps > 0 and world == 1 and not grad:
            out["cpu_baseline"] = cpu_baseline(w, args.cpu_steps)
        print(json.dumps(out))
    e.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
