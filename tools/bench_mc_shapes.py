#!/usr/bin/env python3
"""Scoring-kernel time over problem shapes (k traces, n components, t samples), 2^20 samples, next to the
fp64 VALU issue bound at 2.0 GHz (the clock the chip sustains under this load).  n = 12 takes the generic
moment kernel, everything else (n <= 9) the lane-per-sample kernel."""
import sys, numpy as np
sys.path.insert(0, '.')
from full_waveform_inversion_amd import source_inversion as si
rng = np.random.default_rng(0)
N = 1 << 20
for (k, n, t) in [(21, 9, 512), (21, 6, 512), (21, 3, 512), (21, 9, 100), (21, 6, 100), (21, 3, 100), (5, 6, 2048), (21, 5, 512), (21, 12, 512)]:
    G = rng.standard_normal((k, n, t)); Ms = rng.standard_normal((n, N)); d = np.einsum("kjt,j->kt", G, Ms[:, 5])
    si.score_samples(d, G, Ms[:, :1024], "VR", False, False)
    for metric in ("VR", "CC-shift"):
        ms = min(si.score_samples(d, G, Ms, metric, False, False, return_timing=True)[3] for _ in range(3))
        valu = k * t * (n + (4.5 if metric == "VR" else 8)) * N / 64 * 4 / 1024  # cycles per SIMD
        print("k=%d n=%d t=%d %-8s %.2f ms  %.0f M samples/s  VALU-bound@2.0GHz %.2f ms" % (k, n, t, metric, ms, N / ms / 1e3, valu / 2.0e6), flush=True)
