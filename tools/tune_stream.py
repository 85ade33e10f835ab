#!/usr/bin/env python3
"""Sweep the stencil kernels' launch parameters on one GPU (interleaved rounds, one process).

Prints Gpts/s and the algorithmic-bandwidth fraction per variant; used to pick
the defaults in fwi_kernels.hip::stream_default_tuning.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from full_waveform_inversion_amd import Engine, workloads  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--nt", type=int, default=200)
    ap.add_argument("--npml", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--ty", default="4,8")
    ap.add_argument("--zchunk", default="16,32,64,128,256")
    ap.add_argument("--pf", default="2", help="prefetch depths to sweep (1..3)")
    ap.add_argument("--point", type=int, default=1)
    ap.add_argument("--mode", default="forward", choices=["forward", "save", "adjoint"],
                    help="forward: plain steps; save: forward storing the imaging term; "
                         "adjoint: reverse-time steps with the imaging condition")
    a = ap.parse_args()
    w = workloads.cfg4(a.grid / 256.0, npml=a.npml)
    w.nt = a.nt
    wav = w.wavelet()
    model = w.c.astype(np.float32)
    variants = [("point", 0, 0, 0)] if a.point else []
    for pf in map(int, a.pf.split(",")):
        for ty in map(int, a.ty.split(",")):
            for zc in map(int, a.zchunk.split(",")):
                if zc <= a.grid:
                    variants.append(("stream", ty, zc, pf))
    engines = []
    for k, ty, zc, pf in variants:
        if ty:
            os.environ["FWI_STREAM_TY"] = str(ty)
            os.environ["FWI_STREAM_PF"] = str(pf)
        e = Engine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml, kernel=k, zchunk=zc)
        e.set_model(model)
        engines.append(e)
    res = {v: [] for v in variants}
    ref = None
    for r in range(a.rounds + 1):
        for v, e in zip(variants, engines):
            d = e.forward(None, (w.src_idx, wav), w.rec_idx, save=a.mode != "forward")
            ms = e.last_loop_ms()
            if a.mode == "adjoint":
                e.adjoint(np.ones_like(d) * 1e-3)
                ms = e.last_loop_ms()
            if ref is None:
                ref = d
            err = float(np.linalg.norm(d - ref) / max(np.linalg.norm(ref), 1e-30))
            if r:
                res[v].append(ms)
            if err > 1e-5:
                print("MISMATCH", v, err)
    npts = int(np.prod(w.shape))
    bpp = {"forward": 16, "save": 20, "adjoint": 24}[a.mode]  # bytes/update (adjoint: paired imaging; SURVEY s.8d: 28 unpaired)
    print("grid %d^3 nt %d npml %d mode %s (%d B/update)" % (a.grid, a.nt, a.npml, a.mode, bpp))
    print("%-8s %3s %6s %2s %10s %10s %8s %8s" % ("kernel", "ty", "zchunk", "pf", "us/step", "Gpts/s", "GB/s", "frac8T"))
    for v in variants:
        ms = float(np.median(res[v]))
        us = 1e3 * ms / a.nt
        g = npts / us / 1e3
        print("%-8s %3d %6d %2d %10.2f %10.1f %8.0f %8.3f" % (v[0], v[1], v[2], v[3], us, g, bpp * g, bpp * g / 8000))


if __name__ == "__main__":
    main()
