#!/usr/bin/env python3
"""Per-time-step device time of any BASELINE config on one GPU (forward / save / adjoint)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from full_waveform_inversion_amd import Engine, workloads  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--nt", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--npml", type=int, default=-1, help="override the workload's border width")
    ap.add_argument("--kernel", default="auto")
    ap.add_argument("--zchunk", type=int, default=0, help="planes per workgroup of the 3-D stream kernel (0 = tuned)")
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--abc", default="sponge", choices=["sponge", "cpml"])
    ap.add_argument("--update-form", default="standard", choices=["standard", "increment"])
    ap.add_argument("--store-dtype", default="native", choices=["native", "bf16"])
    ap.add_argument("--launch-mode", default="auto", choices=["auto", "stream", "graph"])
    a = ap.parse_args()
    w = workloads.CONFIGS[a.config](a.scale)
    if a.nt:
        w.nt = a.nt
    if a.npml >= 0:
        w.npml = a.npml
    wav = w.wavelet(np.dtype(a.dtype).type)
    src = w.src_idx[:1]
    e = Engine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml, kernel=a.kernel, dtype=a.dtype, abc=a.abc,
               update_form=a.update_form, store_dtype=a.store_dtype, zchunk=a.zchunk, launch_mode=a.launch_mode,
               pml_alpha_max=(3.14159 * w.f0 if a.abc == "cpml" else 0.0))
    e.set_model(w.c.astype(a.dtype))
    npts = int(np.prod(w.shape))
    out = {}
    for r in range(a.rounds + 1):
        e.forward(None, (src, wav), w.rec_idx, save=False)
        t_f = e.last_loop_ms()
        d = e.forward(None, (src, wav), w.rec_idx, save=True)
        t_s = e.last_loop_ms()
        e.adjoint(d)
        t_a = e.last_loop_ms()
        if r:
            for k, v in (("forward", t_f), ("save", t_s), ("adjoint", t_a)):
                out.setdefault(k, []).append(v)
    print("%s %s nt=%d order=%d npml=%d kernel=%s dtype=%s abc=%s update_form=%s store_dtype=%s launch_mode=%s" % (
        w.name, w.shape, w.nt, w.order, w.npml, e.kernel_name, a.dtype, a.abc, a.update_form, a.store_dtype, a.launch_mode))
    for k, bpp in (("forward", 16), ("save", 20), ("adjoint", 24)):  # adjoint: paired imaging
        us = 1e3 * float(np.median(out[k])) / w.nt
        g = npts / us / 1e3
        print("  %-8s %8.2f us/step %8.1f Gpts/s  %6.0f GB/s algorithmic (%d B/update x%s)" % (
            k, us, g, bpp * g * (2 if a.dtype == "float64" else 1), bpp, "2 fp64" if a.dtype == "float64" else "1"))


if __name__ == "__main__":
    main()
