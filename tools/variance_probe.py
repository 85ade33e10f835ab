#!/usr/bin/env python3
"""Where does the run-to-run spread of an HBM-regime leg come from?  One process, the 256^3 CPML forward sweep timed
`--shots` times on each of `--contexts` freshly created contexts (new device allocations each time).  A spread between
contexts but not within one says buffer placement; a spread within one context says clocks / temperature."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from full_waveform_inversion_amd import Engine, workloads  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--contexts", type=int, default=6)
    ap.add_argument("--shots", type=int, default=4)
    ap.add_argument("--nt", type=int, default=300)
    ap.add_argument("--abc", default="cpml")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--update-form", default="standard")
    ap.add_argument("--zero-wavelet", action="store_true", help="source amplitude 0: the sweep moves all-zero fields")
    a = ap.parse_args()
    w = workloads.cfg4(a.scale, npml=16)
    w.nt = a.nt
    wav = w.wavelet(np.float32)
    if a.zero_wavelet:
        wav = np.zeros_like(wav)
    rows = []
    for i in range(a.contexts):
        e = Engine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml, abc=a.abc, update_form=a.update_form,
                   pml_alpha_max=(np.pi * 10.0 if a.abc == "cpml" else 0.0))
        e.set_model(w.c.astype(np.float32))
        t = []
        for r in range(a.shots):
            e.forward(None, (w.src_idx[:1], wav), w.rec_idx, save=False)
            t.append(round(1e3 * e.last_loop_ms() / w.nt, 2))
        pl = e.placement_info() if hasattr(e, "placement_info") else None
        e.close()
        rows.append(t)
        print("context %d: us/step per shot %s%s" % (i, t, "" if not pl or not pl[0] else
              "  placement search %.1f -> %.1f us, offsets MiB %s" % (pl[0], pl[1], [s >> 20 for s in pl[2][:5]])), flush=True)
    print(json.dumps({"probe": "variance", "abc": a.abc, "shape": list(w.shape), "us_per_step": rows}))


if __name__ == "__main__":
    main()
