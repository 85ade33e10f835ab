#!/usr/bin/env python3
"""The reference's job -- Monte Carlo moment-tensor inversion -- through its own function names on one MI355X:

    python examples/source_inversion_demo.py [--samples 1000000] [--metric VR]

Synthetic traces for a known double-couple source are written as the text files the reference reads, then
`reference_api.run(...)` is called with the reference's positional arguments: trace loading, least-squares estimate,
10^6 random sources scored on the GPU, result files in the reference's layout.
"""
import argparse
import os
import pickle
import random
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import full_waveform_inversion_amd.reference_api as full_waveform_inversion  # noqa: E402

HYP = """NLLOC "demo" "LOCATED" "Location completed."
GEOGRAPHIC  OT 2014 06 29  18 42   10.123456  Lat -75.1 Long -84.2 Depth 2.0
PHASE ID Ins Cmp On Pha  FM Date     HrMn   Sec     Err  ErrMag    Coda      Amp       Per  >   TTpred    Res       Weight    StaLoc(X  Y         Z)        SDist    SAzim  RAz  RDip RQual    Tcorr
ST01   ?    ?    ? P      ? 20140629 1842   10.5000 GAU  2.00e-02 -1.00e+00 -1.00e+00 -1.00e+00 >     0.3770  0.0000    1.0000    1.0000    2.0000    0.0000    0.8000 110.00 110.0  95.0  9     0.0000
ST02   ?    ?    ? P      ? 20140629 1842   10.6000 GAU  2.00e-02 -1.00e+00 -1.00e+00 -1.00e+00 >     0.4770  0.0000    1.0000    3.0000    1.0000    0.0000    0.9000 250.00 250.5 100.0  9     0.0000
END_PHASE
END_NLLOC
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=1000000)
    ap.add_argument("--metric", default="VR", choices=["VR", "CC", "PCC", "CC-shift", "gau"])
    a = ap.parse_args()
    rng = np.random.default_rng(1)
    k, t = 21, 512                                   # the reference's shipped shape: 21 traces
    ker = np.hanning(9) / np.hanning(9).sum()
    G = np.apply_along_axis(lambda v: np.convolve(v, ker, mode="same"), 2, rng.standard_normal((k, 6, t)))
    G *= np.exp(-np.arange(t) / (0.5 * t)) * 1e-10
    np.random.seed(7)
    random.seed(7)
    M_true = full_waveform_inversion.generate_random_DC_MT()               # (6, 1), unit norm
    d = np.einsum("kjt,j->kt", G * 1e10, M_true[:, 0])                     # (loader applies the 1e3 * 1e7 unit factors)
    d += 0.02 * np.abs(d).max() * rng.standard_normal(d.shape)
    with tempfile.TemporaryDirectory() as tmp:
        real, mt = [], []
        for i in range(k):
            real.append("real_%02d.txt" % i)
            mt.append("gf_mt_%02d.txt" % i)
            np.savetxt(os.path.join(tmp, real[-1]), d[i])
            np.savetxt(os.path.join(tmp, mt[-1]), G[i].T)
        hyp = os.path.join(tmp, "event.hyp")
        open(hyp, "w").write(HYP)
        out = os.path.join(tmp, "out")
        labels = ["ST%02d, Z" % i for i in range(k)]
        t0 = time.perf_counter()
        MTs, MTp, MTp_abs = full_waveform_inversion.run(
            tmp, out, real, mt, [], labels, "DC", False, False, a.samples, a.metric, [], [], hyp,
            return_absolute_similarity_values_switch=True)
        el = time.perf_counter() - t0
        res = pickle.load(open(os.path.join(out, "20140629184210123456_FW_DC.pkl"), "rb"))
    best = MTs[:, int(np.argmax(MTp))]
    best = best / np.linalg.norm(best)
    cosang = abs(float(best @ M_true[:, 0]))
    print("%d samples of a double couple against %d traces x %d samples (%s): %.2f s end to end, files: %s"
          % (a.samples, k, t, a.metric, el, sorted(res)))
    print("best sample vs true source: |cos angle| = %.4f; posterior of the best = %.3e" % (cosang, MTp.max()))
    return 0 if cosang > 0.9 else 1


if __name__ == "__main__":
    sys.exit(main())
