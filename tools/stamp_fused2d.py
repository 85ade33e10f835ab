#!/usr/bin/env python3
"""Where the time of one step2d_fused launch goes: s_memtime stamps at the phase boundaries of every workgroup.

Needs the diagnostic build (never the production library):
    make -C full_waveform_inversion_amd/csrc stamps
    FWI_HIP_LIB=full_waveform_inversion_amd/libfwi_hip_stamps.so python tools/stamp_fused2d.py [--mode forward|save|adjoint]
Prints the median over workgroups of each phase in shader cycles and in ns (cycles / the clock read off
s_memrealtime against the HIP-event duration is not needed: the last launch's stamps are differences of s_memtime,
converted with the measured us per launch)."""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from full_waveform_inversion_amd import Engine, _lib, workloads  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="forward", choices=["forward", "save", "adjoint"])
    ap.add_argument("--nt", type=int, default=400)
    a = ap.parse_args()
    w = workloads.cfg2(1.0)
    w.nt = a.nt
    wav = w.wavelet()
    with Engine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml) as e:
        e.set_model(w.c.astype(np.float32))
        for _ in range(2):
            d = e.forward(None, (w.src_idx[:1], wav), w.rec_idx, save=a.mode != "forward")
            ms = e.last_loop_ms()
            if a.mode == "adjoint":
                e.adjoint(d)
                ms = e.last_loop_ms()
        lib = _lib.load()
        ntile = (w.shape[0] // 64) * (w.shape[1] // 64)
        buf = (C.c_ulonglong * (ntile * 8))()
        rc = lib.fwi_debug_fused2d_stamps(buf, ntile * 8)
        assert rc == 0, "not the stamped build? set FWI_HIP_LIB"
    st = np.array(buf, dtype=np.uint64).reshape(ntile, 8).astype(np.int64)
    names = ["fill (LDS-DMA) + barrier", "substep 0", "substep 1", "substep 2", "substep 3", "final stores"]
    d = np.diff(st[:, :7], axis=1)
    tot = st[:, 6] - st[:, 0]
    us_launch = 1e3 * ms / (w.nt / 4)
    out = {"mode": a.mode, "us_per_launch_hip_events": round(us_launch, 3),
           "workgroup_lifetime_cycles_median": float(np.median(tot)),
           "phases_cycles_median": {n: float(np.median(d[:, i])) for i, n in enumerate(names)},
           "phases_fraction_of_lifetime": {n: round(float(np.median(d[:, i]) / np.median(tot)), 3)
                                           for i, n in enumerate(names)},
           "start_skew_cycles_p10_p90": [float(np.percentile(st[:, 0] - st[:, 0].min(), 10)),
                                          float(np.percentile(st[:, 0] - st[:, 0].min(), 90))]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
