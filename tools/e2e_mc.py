#!/usr/bin/env python3
"""End-to-end (PCIe- and host-inclusive) wall time of the Monte Carlo source inversion, 2^20 samples of
the reference's shipped shape (k = 21, n = 9, t = 512), next to the kernel time:
  score   - fwi_mc_score with samples already drawn on the host (75 MB up, 16 MB down)
  host    - samplers.draw on the host (NumPy, batched) + fwi_mc_score
  invert  - fwi_mc_invert: samples drawn on the device, everything returned (75 MB + 24 MB down)
  scores  - fwi_mc_invert with return_samples=False (only 16 MB of scores cross PCIe)
  plan    - the same through fwi_mc_plan_invert, 16 blocks back to back (buffers and Green's functions resident)
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from full_waveform_inversion_amd import samplers, source_inversion as si  # noqa: E402


def best(f, n=4):
    out = []
    for _ in range(n):
        t0 = time.perf_counter()
        r = f()
        out.append((time.perf_counter() - t0, r))
    return min(out, key=lambda x: x[0])


def main():
    rng = np.random.default_rng(0)
    k, n, t, N = 21, 9, 512, 1 << 20
    typ = "DC_single_force_no_coupling"
    G = rng.standard_normal((k, n, t))
    Ms = samplers.draw(typ, N, rng)[0]
    d = np.einsum("kjt,j->kt", G, Ms[:, 5])
    si.score_samples(d, G, Ms[:, :1024], "VR", False, False)
    si.invert_on_device(d, G, 1024, typ, comparison_metric="VR")
    rows = {}
    w, r = best(lambda: si.score_samples(d, G, Ms, "VR", False, False, return_timing=True))
    rows["score"] = (w, r[3])
    t0 = time.perf_counter()
    samplers.draw(typ, N, np.random.default_rng(1))
    draw_s = time.perf_counter() - t0
    rows["host"] = (w + draw_s, r[3])
    w, r = best(lambda: si.invert_on_device(d, G, N, typ, 1, 0, 1.0, "VR", False, False, True, 0, True))
    rows["invert"] = (w, r[5])
    w, r = best(lambda: si.invert_on_device(d, G, N, typ, 1, 0, 1.0, "VR", False, False, False, 0, True))
    rows["scores"] = (w, r[5])
    # a run in blocks through one plan: 16 x 2^20 samples, scores only (what monte_carlo_best_of does)
    with si.MonteCarloPlan(d, G, N) as plan:
        plan.invert(typ, N, 1, 0, 1.0, "VR", False, False, return_samples=False)
        t0 = time.perf_counter()
        kms = 0.0
        for b in range(16):
            plan.invert(typ, N, 1, b * N, 1.0, "VR", False, False, return_samples=False)
            kms += plan.last_kernel_ms
        rows["plan x16 (per block)"] = ((time.perf_counter() - t0) / 16, kms / 16)
    for name, (wall, kms) in rows.items():
        print(json.dumps({"path": name, "inversion_type": typ, "samples": N, "wall_ms": round(wall * 1e3, 2),
                          "score_kernel_ms": round(kms, 3), "samples_per_s_end_to_end": round(N / wall)}), flush=True)


if __name__ == "__main__":
    main()
