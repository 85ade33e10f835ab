#!/usr/bin/env python3
"""Forward-step rate across grid sizes on one GPU (3-D cubes and 2-D squares, O(8), sponge border of 16): where the
Infinity-Cache regime ends, and what sizes that are not multiples of the tile cost.  One JSON line per size."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from full_waveform_inversion_amd import Engine  # noqa: E402


def rate(shape, nt, npml, mode, abc="sponge"):
    n = shape[0]
    c = np.full(shape, 2000.0, np.float32)
    w = np.zeros(nt, np.float32)
    w[:8] = 1.0
    src = [[s // 2 for s in shape]]
    rec = [[min(npml + 2, s - 1) for s in shape]]
    with Engine(shape, 10.0, 1e-3, nt, order=8, npml=npml, sigma_max=300.0, abc=abc,
                pml_alpha_max=(30.0 if abc == "cpml" else 0.0)) as e:
        e.set_model(c)
        best = {}
        for _ in range(3):
            d = e.forward(None, (src, w), rec, save=(mode == "gradient"))
            t = {"forward": e.last_loop_ms()}
            if mode == "gradient":
                e.adjoint(d)
                t["adjoint"] = e.last_loop_ms()
            for k, v in t.items():
                best[k] = min(best.get(k, 1e30), v)
        npts = float(np.prod(shape))
        out = {"shape": list(shape), "kernel": e.kernel_name, "nt": nt, "npml": npml, "abc": abc}
        for k, v in best.items():
            us = 1e3 * v / nt
            out[k + "_us_per_step"] = round(us, 3)
            out[k + "_Gpts_per_s"] = round(npts / us / 1e3, 1)
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ndim", type=int, default=3)
    ap.add_argument("--sizes", default="")
    ap.add_argument("--shapes", default="", help="explicit shapes instead of cubes / squares, e.g. 200x400x400,128x512x512")
    ap.add_argument("--npml", type=int, default=16)
    ap.add_argument("--mode", default="forward", choices=["forward", "gradient"])
    ap.add_argument("--abc", default="sponge", choices=["sponge", "cpml"])
    ap.add_argument("--nt", type=int, default=0, help="time steps per run (0 = by size, 16 .. 400: short runs see a grid that is "
                    "still mostly zeros and read a few % high, DESIGN.md s.4; 1000+ for figures to quote)")
    a = ap.parse_args()
    if a.sizes:
        sizes = [int(s) for s in a.sizes.split(",")]
    elif a.ndim == 3:
        sizes = [96, 128, 160, 192, 200, 224, 256, 300, 320, 384, 400, 448, 500, 512, 600, 640, 768]
    else:
        sizes = [256, 384, 500, 512, 768, 1000, 1024, 1500, 2048, 3000, 4096, 8192]
    shapes = [(n,) * a.ndim for n in sizes]
    if a.shapes:
        shapes = [tuple(int(v) for v in sh.split("x")) for sh in a.shapes.split(",")]
    for shape in shapes:
        npts = float(np.prod(shape))
        nt = a.nt if a.nt > 0 else max(16, min(400, int(4e9 / npts))) // 4 * 4
        if a.mode == "gradient":  # the forward-term store must fit
            nt = max(8, min(nt, int(100e9 / (4.0 * npts)))) // 4 * 4
        print(json.dumps(rate(shape, nt, min(a.npml, min(shape) // 4), a.mode, a.abc)), flush=True)


if __name__ == "__main__":
    main()
