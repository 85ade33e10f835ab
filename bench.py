#!/usr/bin/env python3
"""Headline benchmark: grid-point updates/s of the 3-D 256^3 O(8) acoustic stencil.

One "step" = one full shot of BASELINE.json configs[3] (256^3 constant velocity, 1000 time steps, O(8),
fp32) through fwi_forward() on one MI355X.  With N > 1 (launched by torch.distributed.run, one process
per GPU) every rank runs its own shot per step -- shots are independent -- and the K steps of the timed
region form one gradient evaluation over N*K shots, closed by the path's one real exchange: ONE RCCL
all-reduce of the model-sized gradient accumulator ("an RCCL all-reduce ... only for the final gradient
sum", north_star), inside the timed region.  The reported value is the aggregate over ranks.

No torch in this process: the launcher only provides RANK / WORLD_SIZE / MASTER_*; the control plane
(unique-id hand-off, barriers) is the package's stdlib TCP rendezvous, the data path is HIP + RCCL behind
the C-ABI, and the max-over-ranks of the elapsed time goes through RCCL too (fwi_allreduce_f64_max).
With N > 1 any RCCL failure is fatal (non-zero exit): there is no host-side fallback for the sum.

At N = 1 the same invocation also runs short extra legs and reports them in the same JSON line (`legs`, and
`roofline.hbm_regime`): the 512^3 forward run (working set past the 256 MiB Infinity Cache: the honest HBM
number), the 256^3 gradient shot (forward + store, adjoint + imaging) in the standard and in the increment update
form (the fp32 mode that meets north_star's 1e-5 end to end), configs[1] (2-D 1024^2 x 2000 steps) with the sponge
and with the convolutional PML, and the 256^3 / npml 16 run with the convolutional PML.  `--leg NAME` runs one leg
alone (what the rocprofv3 passes use).

For every N (also N = 1 under FWI_BENCH_FORCE_EXCHANGE=1) `legs.gradient` is the path's real multi-GPU pattern
(configs[2] / configs[4]): per shot forward + store and adjoint + imaging into the device accumulator, then the ONE
RCCL all-reduce of that accumulator, with a check in the line: g.g after the sum = N^2 x g.g before it (every rank
runs the same shot, so the sum is N g).
"""
import argparse
import faulthandler
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0        # same guide: measured float4 copy, the achievable ceiling
BYTES_PER_UPDATE = 16        # read u_cur, u_prev, C; write u_next (fp32)
MALL_BYTES = 256 << 20       # Infinity Cache


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
    out = {"value": round(int(np.prod(w.shape)) * sample_steps / el / 1e9, 3), "unit": "Gpts/s", "cores": cores,
           "kind": "port",
           "sample": "same %s grid, first %d of %d time steps, fp64 OpenMP C port of the build's oracle "
                     "(the reference has no such path), %.1f s" % ("x".join(map(str, w.shape)),
                                                                   sample_steps, w.nt, el)}
    # the NumPy form of the same oracle (what a "NumPy CPU path" of this scheme is): sliced-array stencil,
    # effectively one core; a few steps of the same grid are enough for a rate
    from oracle import fwi_oracle as fo
    q = fo.Propagator(w.c, w.h, w.dt, w.order, w.npml)
    nsteps = 4
    t0 = time.perf_counter()
    q.forward(w.src_idx, wav[:nsteps], w.rec_idx, save=False)
    el_np = time.perf_counter() - t0
    out["numpy_1core"] = {"value": round(int(np.prod(w.shape)) * nsteps / el_np / 1e9, 4), "unit": "Gpts/s",
                          "cores": 1,
                          "sample": "%d time steps of the NumPy oracle on the same grid, %.1f s" % (nsteps, el_np)}
    return out


def measured_traffic(leg):
    """Fabric-side bytes per launch of `leg`'s dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r04_traffic.json, else r03 / r02, written by tools/summarize_profile.py from separate --pmc FETCH_SIZE /
    WRITE_SIZE runs of `bench.py --leg <leg>`), or None.  PMC counters cannot be collected in-process."""
    for name in ("r04_traffic.json", "r03_traffic.json", "r02_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        try:
            ent = json.load(open(path)).get(leg)
        except (OSError, ValueError):
            continue
        if ent:
            return ent.get("traffic_bytes_per_launch"), os.path.relpath(path, ROOT) + "#" + leg
    return None, None


def run_leg(make_workload, device, steps, warmup, grad=False, engine_kw=None, exchange=None, barrier=None,
            max_over_ranks=None, dtype=None):
    """Time `steps` shots of a workload; returns (workload, per-shot wall s, kernel us/launch, info)."""
    from full_waveform_inversion_amd import Engine
    w = make_workload()
    ft = np.float64 if dtype == "float64" else np.float32
    wav = w.wavelet(ft)
    model = w.c.astype(ft)
    kw = dict(order=w.order, npml=w.npml, device=device)
    kw.update(engine_kw or {})
    e = Engine(w.shape, w.h, w.dt, w.nt, **kw)
    try:
        t_sm = time.perf_counter()
        e.set_model(model)
        set_model_ms = 1e3 * (time.perf_counter() - t_sm)
        ex = exchange(e) if exchange else None

        def step():
            d = e.forward(None, (w.src_idx[:1], wav), w.rec_idx, save=grad)
            ms = e.last_loop_ms()
            if grad:
                e.adjoint(d)  # residual = the data themselves: same work as any residual
                ms += e.last_loop_ms()
            return ms

        for _ in range(warmup):
            step()
        if ex is not None and grad:
            e.allreduce_gradient()  # warm the communicator (the first collective sets up the rings)
        check = None
        if grad and ex is not None:
            e.vec_create(1)  # scratch for the check below (outside the timed region)
        e.reset_gradient()
        if barrier:
            barrier()
        e.synchronize()
        t0 = time.perf_counter()
        loop_ms = [step() for _ in range(steps)]
        e.synchronize()
        t_shots = time.perf_counter() - t0
        if grad and ex is not None:  # untimed: g.g of this rank's accumulated gradient before the sum
            e.gradient_vec(0, "slowness2")
            gg_before = e.vec_dot(0, 0)
            e.synchronize()
        t1 = time.perf_counter()
        if ex is not None and grad:
            e.allreduce_gradient()  # the shot loop's one exchange: sum of the per-rank gradients
        e.synchronize()
        # (the collective alone: taken BEFORE the control-plane barrier that closes the timed region)
        allreduce_ms = 1e3 * (time.perf_counter() - t1) if (ex is not None and grad) else None
        if barrier:
            barrier()
        el = t_shots + (time.perf_counter() - t1)
        if grad and ex is not None:
            e.gradient_vec(0, "slowness2")
            gg_after = e.vec_dot(0, 0)
            nr = ex.rccl_ranks
            # (FWI_BENCH_CHECK_EXPECT_RANKS: the rehearsal of a FAILED check on a one-GPU box -- the test pretends the
            # sum should have come from that many ranks, tests/test_gpu_bench_contract.py)
            nr = int(os.environ.get("FWI_BENCH_CHECK_EXPECT_RANKS", nr))
            ratio = gg_after / gg_before if gg_before > 0 else float("nan")
            check = {"g_dot_g_before_sum": gg_before, "g_dot_g_after_sum": gg_after, "ratio": ratio,
                     "expected_ratio": float(nr * nr), "ok": bool(abs(ratio - nr * nr) <= 1e-4 * nr * nr),
                     "note": "every rank runs the same shots, so the all-reduced accumulator is N x the local one"}
            # A wrong or skipped RCCL sum must not exit 0 with a headline figure: the ranks agree on the
            # outcome (the minimum of their verdicts, over RCCL itself) and ALL of them end non-zero.  At N = 1 the
            # expected ratio is 1 whether or not the all-reduce ran -- only N > 1 discriminates, and no N > 1 run has
            # been possible on the one-GPU boxes this was built on.
            any_bad = e.allreduce_f64([0.0 if check["ok"] else 1.0], op="max")[0] > 0.0
            if any_bad:
                print("bench.py: gradient all-reduce check FAILED on %s: g.g after / before the sum = %r, expected %d"
                      % ("this rank" if not check["ok"] else "another rank", ratio, nr * nr), file=sys.stderr)
                raise SystemExit(3)
        if max_over_ranks:
            el = max_over_ranks(e, el)
        sweeps = 2 if grad else 1
        info = {"kernel": e.kernel_name, "set_model_ms": round(set_model_ms, 1),
                "rccl_ranks": ex.rccl_ranks if ex is not None else None, "allreduce_check": check,
                "allreduce_ms": None if allreduce_ms is None else round(allreduce_ms, 3)}
        pl = e.placement_info()
        if pl[0]:  # the context placed some of its arrays by measurement (3-D CPML / increment form; fwi_placement_info)
            info["placement"] = {"us_before": round(pl[0], 2), "us_after": round(pl[1], 2),
                                 "offsets_MiB": [s >> 20 for s in pl[2][:5]]}
        return w, el, 1e3 * float(np.mean(loop_ms)) / (w.nt * sweeps), info
    finally:
        e.close()


def roofline_entry(leg, w, kern_us, bpu, bound, note, steps_per_launch=1, extra_bytes=0):
    """`kern_us`: HIP-event time of the step loop / launches; one launch advances `steps_per_launch` time steps.
    `extra_bytes`: algorithmic bytes per launch beside the per-update figure (the CPML's memory variables)."""
    npts = int(np.prod(w.shape)) * steps_per_launch
    achieved = (bpu * npts + extra_bytes) / (kern_us * 1e-6) / 1e9
    traffic, src = measured_traffic(leg)
    return {"bound": bound, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "frac_of_achievable_hbm": round(achieved / HBM_COPY_GBS, 4),
            "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": src,
            "algorithmic_bytes_per_launch": bpu * npts + extra_bytes, "kernel_avg_us": round(kern_us, 3), "note": note}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--grid", type=int, default=256, help="cube edge of the headline leg (256 = BASELINE config)")
    ap.add_argument("--nt", type=int, default=1000)
    ap.add_argument("--npml", type=int, default=0)
    ap.add_argument("--kernel", default="auto")
    ap.add_argument("--zchunk", type=int, default=0)
    ap.add_argument("--image-stride", type=int, default=1,
                    help="gradient mode: store / correlate the forward term every S-th step (fwi_config.image_stride)")
    ap.add_argument("--mode", default="forward", choices=["forward", "gradient"],
                    help="headline leg: forward = the stencil run; gradient = forward(save) + adjoint(imaging) per step")
    ap.add_argument("--leg", default="all", choices=["all", "headline", "hbm", "gradient", "gradient_increment", "cfg2",
                                                     "cfg2_cpml", "cpml3d", "cpml3d_adjoint", "cpml512", "fp64", "point", "bf16"],
                    help="all: headline + (at N = 1) the three extra legs; or one leg alone (profiling passes)")
    ap.add_argument("--leg-nt", type=int, default=0, help="time steps of a --leg run (0 = the leg's own)")
    ap.add_argument("--cpu-steps", type=int, default=200,
                    help="time steps of the CPU baseline sample (200 of the 1000 steps, ~3 s on 16 cores); 0 = skip")
    args = ap.parse_args()

    # one node, one process per GPU: the driver only supports dmabuf IPC, and the RCCL bootstrap of a
    # single-node job belongs on loopback (the container's hostname may not resolve)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    from full_waveform_inversion_amd import _lib as _fl
    ndev = _fl.device_count()  # a launcher may have narrowed this process to its own GPU: then ordinal 0
    if ndev < 1:
        sys.exit("bench.py: no HIP device visible (there is no CPU fallback)")
    local %= ndev
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run "
                     "--nproc-per-node %d" % (args.gpus, args.gpus))
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    if world > 1 and world > ndev and "LOCAL_RANK" in os.environ and ndev > 1:
        sys.exit("bench.py: %d ranks but only %d visible GPUs (RCCL refuses two ranks on one device)" % (world, ndev))

    from full_waveform_inversion_amd import workloads
    from full_waveform_inversion_amd.rendezvous import Rendezvous, RendezvousError
    from full_waveform_inversion_amd.shots import RcclExchange

    # FWI_BENCH_FORCE_EXCHANGE=1: take the N > 1 code path (rendezvous, RCCL communicator, timed all-reduce, max over
    # ranks) even with one rank -- how the one-GPU test box rehearses exactly what the driver launches for N = 2, 4, 8
    multi = world > 1 or os.environ.get("FWI_BENCH_FORCE_EXCHANGE") == "1"
    rdzv = Rendezvous.from_env(timeout=float(os.environ.get("FWI_RDZV_TIMEOUT", "300"))) if multi else None

    def exchange(e):
        # a communicator that neither comes up nor fails must not hang the job: hard exit with a traceback
        faulthandler.dump_traceback_later(float(os.environ.get("FWI_COMM_INIT_TIMEOUT", "300")), exit=True)
        # RCCL prints a version banner on STDOUT at communicator init (RCCL 2.27: "RCCL version : ...", five lines):
        # this process' stdout carries ONE json line, so the library writes to stderr while it comes up
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            return RcclExchange(e, rdzv)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
            faulthandler.cancel_dump_traceback_later()

    def max_over_ranks(e, el):
        return e.allreduce_f64([el], op="max")[0]

    grad = args.mode == "gradient"
    S = max(1, args.image_stride)

    def headline_workload():
        w = workloads.cfg4(args.grid / 256.0, npml=args.npml)
        w.nt = args.nt
        return w

    out = None
    if args.leg in ("all", "headline"):
        w, el, kern_us, info = run_leg(
            headline_workload, local, args.steps, args.warmup, grad=grad,
            engine_kw=dict(kernel=args.kernel, zchunk=args.zchunk, image_stride=S),
            exchange=exchange if multi else None, barrier=rdzv.barrier if rdzv else None,
            max_over_ranks=max_over_ranks if multi else None)
        if rank == 0:
            npts = int(np.prod(w.shape))
            sweeps = 2 if grad else 1            # forward + adjoint sweep per shot
            # forward+save 20 B; adjoint + paired imaging 24 B (SURVEY s.8d prices the unpaired form at 28 B)
            if grad and S > 1:  # every S-th step stores q (+4 B) / reads q and read-modify-writes g (+12 B)
                bpu = ((16 + 4.0 / S) + (16 + 12.0 / S)) / 2.0
            else:
                bpu = (20 + 24) / 2.0 if grad else BYTES_PER_UPDATE
            value = world * args.steps * npts * w.nt * sweeps / el / 1e9
            resident = 3 * npts * 4 <= MALL_BYTES and not grad
            out = {
                "metric": "stencil grid-point-updates/sec (Gpts/s), 3-D O(8) acoustic" +
                          (" (forward+adjoint sweeps)" if grad else ""),
                "value": round(value, 3), "unit": "Gpts/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(1e3 * el / args.steps, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                "data": "synthetic", "set_model_ms": info["set_model_ms"],
                "config": {"workload": "configs[3]: 3-D %s constant velocity, 1 shot/GPU/step, %d time steps, "
                                       "O(8), npml=%d" % ("x".join(map(str, w.shape)), w.nt, w.npml),
                           "kernel": info["kernel"], "image_stride": S,
                           "parallelism": "shot-parallel x%d" % world,
                           "exchange": "none" if not multi else (
                                       "one rccl allreduce of the gradient accumulator after the K shots (timed)" if grad
                                       else "none in this forward-only leg (shots are independent; the slowest rank's "
                                            "time is taken with a 1-double rccl allreduce); the gradient all-reduce of "
                                            "configs[2] / configs[4] is timed in legs.gradient"),
                           "rccl_ranks": info["rccl_ranks"], "control_plane": "stdlib tcp rendezvous (no torch)"},
                "roofline": roofline_entry(
                    "headline" if (args.grid == 256 and not grad) else "other", w, kern_us, bpu,
                    "infinity-cache-resident" if resident else "hbm",
                    "algorithmic %g B/update x %d updates per launch / HIP-event time of the %d-launch loop(s); "
                    "wavefield working set 3 x %d MiB%s" % (
                        bpu, npts, w.nt * sweeps, npts * 4 >> 20,
                        " fits the 256 MiB Infinity Cache: the fraction of the 8 TB/s HBM peak is kept for "
                        "continuity, the HBM-regime number is roofline.hbm_regime" if resident else "")),
            }

    extra = world == 1 and rank == 0 and not grad and args.grid == 256
    legs = {}
    want = (lambda name: args.leg == name or (args.leg == "all" and extra))

    if want("hbm"):
        def wl():
            w = workloads.cfg4(2.0)
            w.nt = args.leg_nt or 300
            return w
        w, el, kern_us, info = run_leg(wl, local, 3, 1)
        r = roofline_entry("hbm", w, kern_us, BYTES_PER_UPDATE, "hbm",
                           "512^3 forward, %d steps x 3 shots: working set 3 x 512 MiB, past the Infinity Cache" % w.nt)
        r["workload"] = "3-D 512x512x512 constant velocity, O(8), %d time steps" % w.nt
        r["Gpts_per_s"] = round(int(np.prod(w.shape)) / kern_us / 1e3, 1)
        r["kernel"] = info["kernel"]
        legs["hbm"] = r
    npts256 = 256 ** 3

    def gradient_leg(name, engine_kw, bpu, note, shots=3):
        def wl():
            w = workloads.cfg4(1.0)
            w.nt = args.leg_nt or 1000
            return w
        w, el, kern_us, info = run_leg(wl, local, shots, 1, grad=True, engine_kw=engine_kw,
                                       exchange=exchange if multi else None, barrier=rdzv.barrier if rdzv else None,
                                       max_over_ranks=max_over_ranks if multi else None)
        if rank != 0:
            return
        r = roofline_entry(name, w, kern_us, bpu, "hbm", note % (w.nt * int(np.prod(w.shape)) * 4 / 2.0 ** 30))
        r["workload"] = "configs[3] grid, forward(save) + adjoint(imaging), %d time steps, %d shot(s) per GPU, " \
                        "then the gradient all-reduce" % (w.nt, shots)
        r["update_form"] = (engine_kw or {}).get("update_form", "standard")
        r["n_gpus"] = world
        r["Gpts_per_s_both_sweeps"] = round(world * shots * int(np.prod(w.shape)) * w.nt * 2 / el / 1e9, 1)
        r["Gpts_per_s_both_sweeps_kernel_time"] = round(int(np.prod(w.shape)) / kern_us / 1e3, 1)
        r["ms_per_shot_gradient"] = round(1e3 * el / shots, 2)
        r["kernel"] = info["kernel"]
        r["exchange"] = "none (one rank)" if not multi else \
            "one rccl allreduce of the gradient accumulator after the %d shots, inside the timed region" % shots
        r["rccl_ranks"] = info["rccl_ranks"]
        r["allreduce_ms"] = info["allreduce_ms"]
        r["allreduce_check"] = info["allreduce_check"]
        if info.get("placement"):
            r["placement"] = info["placement"]
        legs[name] = r

    # the gradient leg runs on every rank when there is an exchange to time (N > 1, or the forced rehearsal)
    if args.leg == "gradient" or (args.leg == "all" and not grad and args.grid == 256 and (extra or multi)):
        gradient_leg("gradient", None, 22.0,
                     "256^3 gradient shot, STANDARD update form: forward + store (20 B/update), adjoint + paired "
                     "imaging (24 B/update); the forward-term store (%.1f GiB) streams through HBM",
                     shots=2 if multi else 3)
    if want("gradient_increment"):
        gradient_leg("gradient_increment", dict(update_form="increment"), 26.0,
                     "256^3 gradient shot, INCREMENT update form (u, v = u - u_prev; the fp32 mode that meets 1e-5 end "
                     "to end, profiles/r04_parity.json, and what inversions run by default since round 4): forward + store "
                     "24 B/update, adjoint + PAIRED imaging 28 B/update (round 4: the second pairing takes u_prev = u - v; "
                     "32 B unpaired before); four padded fields = 294 MB do not fit the Infinity Cache; store %.1f GiB",
                     shots=2)
    if want("cfg2"):
        def wl():
            w = workloads.cfg2(1.0)
            if args.leg_nt:
                w.nt = args.leg_nt
            return w
        w, el, step_us, info = run_leg(wl, local, 10, 2)
        spl = 4 if info["kernel"] == "step2d_fused" else 1  # FUSED2D_STEPS time steps per launch
        r = roofline_entry("cfg2", w, step_us * spl, BYTES_PER_UPDATE, "lds+valu (the fields are cache-resident: 3 x 4 MiB)",
                           "configs[1]; one launch of the fused kernel advances %d time steps on an LDS-resident tile, "
                           "so it moves FEWER bytes than the per-step algorithmic figure (traffic < algorithmic); the "
                           "fraction of the HBM peak is quoted for continuity only -- DESIGN.md s.4 states the LDS / "
                           "VALU / launch-boundary budget this kernel is measured against" % spl, steps_per_launch=spl)
        r["workload"] = "configs[1]: 2-D 1024x1024 layered, 1 shot, %d steps, O(8) + absorbing border (sponge)" % w.nt
        r["us_per_time_step"] = round(step_us, 3)
        r["Gpts_per_s"] = round(int(np.prod(w.shape)) / step_us / 1e3, 1)
        r["ms_per_shot"] = round(1e3 * el / 10, 3)
        r["kernel"] = info["kernel"]
        legs["cfg2"] = r
    if want("cfg2_cpml"):
        def wl():
            w = workloads.cfg2(1.0)
            if args.leg_nt:
                w.nt = args.leg_nt
            return w
        w, el, step_us, info = run_leg(wl, local, 10, 2, engine_kw=dict(abc="cpml", pml_alpha_max=np.pi * 15.0))
        spl = 4 if info["kernel"] == "step2d_fused" else 1
        mv = 16 * 2 * (2 * w.npml * w.shape[0])  # psi, zeta read + written once per launch, both axes' borders
        r = roofline_entry("cfg2_cpml", w, step_us * spl, BYTES_PER_UPDATE, "lds+valu (cache-resident fields)",
                           "configs[1] with the CONVOLUTIONAL PML (npml %d) carried inside the fused kernel: %d time "
                           "steps per launch, memory variables of the border in LDS beside the tile; the launch ends "
                           "with its four corner tiles, which advance both borders' recursions (DESIGN.md s.4 CPML)"
                           % (w.npml, spl), steps_per_launch=spl, extra_bytes=mv)
        r["workload"] = "configs[1]: 2-D 1024x1024 layered, 1 shot, %d steps, O(8) + CPML npml %d" % (w.nt, w.npml)
        r["us_per_time_step"] = round(step_us, 3)
        r["Gpts_per_s"] = round(int(np.prod(w.shape)) / step_us / 1e3, 1)
        r["ms_per_shot"] = round(1e3 * el / 10, 3)
        r["kernel"] = info["kernel"]
        legs["cfg2_cpml"] = r
    if want("cpml3d"):
        def wl():
            w = workloads.cfg4(1.0, npml=16)
            w.nt = args.leg_nt or 300
            return w
        w, el, step_us, info = run_leg(wl, local, 3, 1, engine_kw=dict(abc="cpml", pml_alpha_max=np.pi * 10.0))
        mv = 16 * 3 * (2 * w.npml * w.shape[0] * w.shape[1])  # psi, zeta of three axes, read + written once per step
        r = roofline_entry("cpml3d", w, step_us, BYTES_PER_UPDATE, "hbm",
                           "256^3 forward with the CONVOLUTIONAL PML, npml %d: fields + 50 MB of memory variables = 271 "
                           "MB, and every byte a step touches counts against the Infinity Cache's 256 MiB (streaming "
                           "hints do not exempt it), so this is an HBM-regime run: algorithmic 369 MB at the ~5.5 TB/s "
                           "the 512^3 run sustains = 67 us is the floor of a fully fused step.  One time step = ONE line "
                           "launch (fwi_pml.hip pml_line_t: the z and y borders' recursions, which hand their term over in "
                           "arrays compact over the border shells) + the step kernel with the x border's recursion in its "
                           "lanes, which adds the handed-over terms inside q (round 4; round 3: the step kernel + one line "
                           "launch per axis that re-read u, C and read-modified-wrote u'), in 8-row tiles x 32 planes -- the "
                           "launch shape of an HBM-regime run; `kernel_avg_us` is the whole step.  Where the step kernel's seven "
                           "arrays lie relative to each other moved this leg between 73 and 85 us from process to process; "
                           "the context now places the four small ones by measurement at creation (`placement`; "
                           "DESIGN.md s.4 CPML)" % w.npml,
                           extra_bytes=mv)
        r["workload"] = "3-D 256x256x256 constant velocity, O(8) + CPML npml %d, %d time steps" % (w.npml, w.nt)
        r["us_per_time_step"] = round(step_us, 3)
        r["Gpts_per_s"] = round(int(np.prod(w.shape)) / step_us / 1e3, 1)
        r["kernel"] = info["kernel"]
        r["placement"] = info.get("placement")
        legs["cpml3d"] = r
    # measurement-only CPML legs (profiles/r04_*; never part of the default line): the gradient sweeps at 256^3 and the
    # forward sweep at 512^3, where stream_default_tuning picks the 8-row tiles
    if args.leg == "cpml3d_adjoint":
        def wl():
            w = workloads.cfg4(1.0, npml=16)
            w.nt = args.leg_nt or 300
            return w
        w, el, step_us, info = run_leg(wl, local, 2, 1, grad=True, engine_kw=dict(abc="cpml", pml_alpha_max=np.pi * 10.0))
        mv = 16 * 3 * (2 * w.npml * w.shape[0] * w.shape[1])
        r = roofline_entry("cpml3d_adjoint", w, step_us, 22.0, "hbm",
                           "256^3 / npml %d gradient shot with the CPML: forward + store (20 B/update) and adjoint + paired "
                           "imaging (24 B/update), one line launch + one step launch per time step either way" % w.npml,
                           extra_bytes=mv)
        r["us_per_time_step_both_sweeps_mean"] = round(step_us, 3)
        r["kernel"] = info["kernel"]
        legs["cpml3d_adjoint"] = r
    if args.leg == "cpml512":
        def wl():
            w = workloads.cfg4(2.0, npml=16)
            w.nt = args.leg_nt or 60
            return w
        w, el, step_us, info = run_leg(wl, local, 2, 1, engine_kw=dict(abc="cpml", pml_alpha_max=np.pi * 10.0))
        mv = 16 * 3 * (2 * w.npml * w.shape[0] * w.shape[1])
        r = roofline_entry("cpml512", w, step_us, BYTES_PER_UPDATE, "hbm",
                           "512^3 forward with the CPML, npml %d: 8-row tiles (two waves per SIMD, 256 registers: the x "
                           "border's variants fit since round 4, no scratch)" % w.npml, extra_bytes=mv)
        r["us_per_time_step"] = round(step_us, 3)
        r["Gpts_per_s"] = round(int(np.prod(w.shape)) / step_us / 1e3, 1)
        r["kernel"] = info["kernel"]
        legs["cpml512"] = r
    for name, kw, bpu, what in (("fp64", dict(dtype="float64"), 32, "fp64 engine (step3d_stream<double>, 32 B/update)"),
                                ("point", dict(kernel="point"), 16, "generic one-thread-per-point kernel (step_point)"),
                                ("bf16", dict(store_dtype="bf16"), None, "bf16 forward-term store")):
        if args.leg != name:
            continue  # measurement-only legs (profiles/r03_*): never part of the default line
        def wl():
            w = workloads.cfg4(1.0)
            w.nt = args.leg_nt or 300
            return w
        if name == "bf16":
            w, el, kern_us, info = run_leg(wl, local, 2, 1, grad=True, engine_kw=kw)
            bpu = (18 + 20) / 2.0
        else:
            w, el, kern_us, info = run_leg(wl, local, 3, 1, engine_kw=kw, dtype=kw.get("dtype"))
        r = roofline_entry(name, w, kern_us, bpu, "hbm", "256^3, %s, %d time steps" % (what, w.nt))
        r["Gpts_per_s"] = round(int(np.prod(w.shape)) / kern_us / 1e3, 1)
        r["kernel"] = info["kernel"]
        legs[name] = r

    # N > 1: the 2-D configuration beside the headline (north_star: "a synthetic 1024^2 / 256^3 grid ... at 1, 2, 4
    # and 8 GPUs"): every rank runs configs[1] shots, barriers and the slowest rank's time go over the control plane.
    # Optional -- a failure here is recorded in the line, never allowed to cost the headline measurement.
    desync = False
    if multi and not grad and not os.environ.get("FWI_BENCH_NO_2D_SCALING") and (
            (args.leg == "all" and args.grid == 256) or os.environ.get("FWI_BENCH_FORCE_2D_SCALING")):
        try:
            rdzv.set_timeout(30.0)
            def wl2():
                w = workloads.cfg2(1.0)
                if args.leg_nt:
                    w.nt = args.leg_nt
                return w
            w2, el2, step_us2, info2 = run_leg(wl2, local, 10, 2, barrier=rdzv.barrier,
                                               max_over_ranks=lambda e, el: rdzv.allreduce([el], op="max")[0])
            if rank == 0:
                npts2 = int(np.prod(w2.shape))
                legs["cfg2"] = {"workload": "configs[1]: 2-D 1024x1024 layered, 1 shot/GPU/step, %d steps, O(8) + "
                                            "absorbing border" % w2.nt, "n_gpus": world, "scaling": "weak",
                                "Gpts_per_s": round(world * 10 * npts2 * w2.nt / el2 / 1e9, 1),
                                "ms_per_shot": round(1e3 * el2 / 10, 3), "us_per_time_step": round(step_us2, 3),
                                "kernel": info2["kernel"], "exchange": "none (forward shots; barriers and the max over "
                                "ranks go over the control plane)"}
        except (RendezvousError, socket.timeout, OSError) as ex:
            # only a control-plane failure is recoverable here (the leg is optional).  An FwiError -- a HIP or RCCL
            # failure on the device -- is NOT caught: the process ends non-zero and the fault is investigated
            desync = True         # a rank may have missed a collective: no further control-plane traffic
            if rank == 0:
                legs["cfg2"] = {"error": "%s: %s" % (type(ex).__name__, ex)}

    if rank == 0:
        if out is None:  # a single extra leg on its own (profiling pass)
            out = {"leg": args.leg}
        if "hbm" in legs and "roofline" in out:
            out["roofline"]["hbm_regime"] = legs.pop("hbm")
        if legs:
            out["legs"] = legs
        if args.cpu_steps > 0 and extra and args.leg == "all":
            out["cpu_baseline"] = cpu_baseline(headline_workload(), args.cpu_steps)
        print(json.dumps(out))
    if rdzv is not None:
        if not desync:
            try:
                rdzv.barrier()
            except Exception:  # noqa: BLE001 -- a peer that failed in the optional leg is already gone
                pass
        rdzv.close()


if __name__ == "__main__":
    main()
