#!/usr/bin/env python3
"""Measured parity of the HIP path against the build's own CPU oracle (relative L2).

Run on the GPU box; the table goes into profiles/ and DESIGN.md.  "e2e" = gradient of
J(model; d_obs) with d_obs given as an input and the residual formed from the GPU's own
forward (fp32 forward error enters the residual, amplified by |d| / |r|); "same-r" = both
paths back-propagate the same residual.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from full_waveform_inversion_amd import Engine, workloads  # noqa: E402
from oracle.c_oracle import CPropagator  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def case(name, w, nt=None, dtype="float32", kernel="auto", npml=None, env=None, **engine_kw):
    for k, v in (env or {}).items():  # creation-time hooks (FWI_STREAM_TY=8: the 8-row tiles of the HBM-regime grids)
        os.environ[k] = v
    try:
        return _case(name, w, nt, dtype, kernel, npml, env, **engine_kw)
    finally:
        for k in (env or {}):
            os.environ.pop(k, None)


def _case(name, w, nt, dtype, kernel, npml, env, **engine_kw):
    if nt:
        w.nt = nt
    if npml is not None:
        w.npml = npml
    okw = {k: engine_kw[k] for k in ("abc", "pml_alpha_max") if k in engine_kw}  # options that change the SCHEME
    wav = w.wavelet(np.float64)
    src = w.src_idx[:1]
    c0 = w.c_init if w.c_init is not None else w.c * (1.0 + 0.03 * np.sin(np.indices(w.shape).sum(0) / 9.0))
    t0 = time.time()
    p = CPropagator(w.c, w.h, w.dt, w.order, w.npml, **okw)
    d_obs = p.forward(src, wav, w.rec_idx, save=False)
    p0 = CPropagator(c0, w.h, w.dt, w.order, w.npml, sigma_max=p.sigma_max, **okw)
    d0 = p0.forward(src, wav, w.rec_idx)
    r = d0 - d_obs
    a0 = p0.adjoint(r)
    g0 = p0.gradient()
    t_cpu = time.time() - t0
    with Engine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml, sigma_max=p.sigma_max, dtype=dtype,
                kernel=kernel, **engine_kw) as e:
        dg = e.forward(c0, (src, wav), w.rec_idx, save=True)
        ag = e.adjoint(r)
        gg = e.gradient()
        e.reset_gradient()
        e.adjoint(dg.astype(np.float64) - d_obs)
        ge = e.gradient()
        kern = e.kernel_name
    out = {"case": name, "shape": list(w.shape), "nt": w.nt, "order": w.order, "npml": w.npml, "dtype": dtype,
           "kernel": kern, "engine_options": engine_kw, "env": env or {}, "seis": rel(dg, d0), "adj_src": rel(ag, a0), "grad_same_r": rel(gg, g0),
           "grad_e2e": rel(ge, g0), "resid_over_data": float(np.linalg.norm(r) / np.linalg.norm(d0)),
           "oracle_seconds": round(t_cpu, 1)}
    print(json.dumps(out), flush=True)
    return out


def main():
    rows = [
        case("cfg1 full", workloads.cfg1(1.0)),
        case("cfg2 256^2 x500", workloads.cfg2(0.25)),
        case("cfg2 512^2 x1000", workloads.cfg2(0.5)),
        case("cfg2 1024^2 x2000 (full)", workloads.cfg2(1.0)),
        case("cfg4 96^3 x375", workloads.cfg4(0.375)),
        case("cfg4 128^3 x500", workloads.cfg4(0.5)),
        case("cfg4 128^3 x500 point", workloads.cfg4(0.5), kernel="point"),
        case("cfg4 128^3 x500 fp64 point", workloads.cfg4(0.5), dtype="float64", kernel="point"),
        case("cfg4 128^3 x500 fp64 stream", workloads.cfg4(0.5), dtype="float64"),
        case("cfg5 128^3 x500 fp64 stream", workloads.cfg5(0.5), dtype="float64"),
        case("cfg5 64^3 x250", workloads.cfg5(0.25)),
        case("cfg5 128^3 x500", workloads.cfg5(0.5)),
        # the fp32 increment-form update (fwi_config.update_form = 1) beside the standard form above
        case("cfg2 512^2 x1000 increment", workloads.cfg2(0.5), update_form="increment"),
        case("cfg2 1024^2 x2000 (full) increment", workloads.cfg2(1.0), update_form="increment"),
        case("cfg4 128^3 x500 increment", workloads.cfg4(0.5), update_form="increment"),
        case("cfg5 64^3 x250 increment", workloads.cfg5(0.25), update_form="increment"),
        case("cfg5 128^3 x500 increment", workloads.cfg5(0.5), update_form="increment"),
        # the convolutional PML (round 3: inside the fused 2-D launch; 3-D: x border in the lanes, z / y line launches)
        case("cfg2 1024^2 x2000 (full) cpml", workloads.cfg2(1.0), abc="cpml", pml_alpha_max=47.0),
        case("cfg2 1000^2 x1000 cpml (seam tiling)", workloads.cfg2(0.9765625), nt=1000, npml=40, abc="cpml", pml_alpha_max=47.0),
        case("cfg5 128^3 x500 cpml npml 16", workloads.cfg5(0.5), npml=16, abc="cpml", pml_alpha_max=31.0),
        case("cfg5 128^3 x500 cpml npml 16 increment", workloads.cfg5(0.5), npml=16, abc="cpml", pml_alpha_max=31.0,
             update_form="increment"),
        case("cfg5 128^3 x500 cpml npml 16 fp64", workloads.cfg5(0.5), npml=16, dtype="float64", abc="cpml", pml_alpha_max=31.0),
        # round 4: the 8-row tiles stream_default_tuning picks for every grid past the Infinity Cache, on oracle-sized grids
        case("cfg5 128^3 x500, 8-row tiles", workloads.cfg5(0.5), env={"FWI_STREAM_TY": "8"}),
        case("cfg5 128^3 x500 increment, 8-row tiles", workloads.cfg5(0.5), update_form="increment", env={"FWI_STREAM_TY": "8"}),
        case("cfg5 128^3 x500 cpml npml 16, 8-row tiles", workloads.cfg5(0.5), npml=16, abc="cpml", pml_alpha_max=31.0,
             env={"FWI_STREAM_TY": "8"}),
        case("cfg5 128^3 x500 cpml npml 16 increment, 8-row tiles", workloads.cfg5(0.5), npml=16, abc="cpml",
             pml_alpha_max=31.0, update_form="increment", env={"FWI_STREAM_TY": "8"}),
        case("cfg5 128^3 x500 fp64 stream, 8-row tiles", workloads.cfg5(0.5), dtype="float64", env={"FWI_STREAM_TY": "8"}),
    ]
    if len(sys.argv) > 1:
        json.dump(rows, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
