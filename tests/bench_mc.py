#!/usr/bin/env python3
"""Throughput of the reference's real hot loop on the GPU (samples/s) next to the pinned CPU oracle.

Shape = the reference's shipped configuration: k = 21 traces, n = 9 components
(full_waveform_inversion.py:48-52), t = 512 samples as in BASELINE.md s.2.
"""
import argparse
import json
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from full_waveform_inversion_amd import source_inversion as si  # noqa: E402
from oracle import mc_oracle as mo  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nsamp", type=int, default=1 << 20)
    ap.add_argument("--cpu-samples", type=int, default=300)
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    k, n, t = 21, 9, 512
    G = rng.standard_normal((k, n, t))
    Mt = rng.standard_normal(n)
    d = np.einsum("kjt,j->kt", G, Mt) + 0.05 * rng.standard_normal((k, t))
    Ms = rng.standard_normal((n, a.nsamp))
    rows = []
    for metric, norm, allat in [("VR", False, False), ("VR", True, True), ("CC", False, False),
                                ("PCC", True, False), ("CC-shift", False, False), ("gau", False, True)]:
        si.score_samples(d, G, Ms[:, :1024], metric, norm, allat)  # warm-up
        best = min(si.score_samples(d, G, Ms, metric, norm, allat, return_timing=True)[3] for _ in range(3))
        t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mo.score_samples(G, d, Ms[:, :a.cpu_samples], metric, norm, allat)
        cpu = a.cpu_samples / (time.perf_counter() - t0)
        rows.append({"metric": metric, "normalise": norm, "all_at_once": allat, "gpu_kernel_ms": round(best, 3),
                     "gpu_samples_per_s": round(a.nsamp / (best * 1e-3)), "cpu_oracle_samples_per_s_1core": round(cpu)})
        print(json.dumps(rows[-1]), flush=True)


if __name__ == "__main__":
    main()
