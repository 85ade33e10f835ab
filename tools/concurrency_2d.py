#!/usr/bin/env python3
"""Do independent 2-D shots overlap on one GPU?  E engines (contexts, streams) driven by E host
threads (ctypes releases the GIL) versus one after the other."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from full_waveform_inversion_amd import Engine, workloads  # noqa: E402


def main():
    # usage: concurrency_2d.py [scale of configs[1], default 1.0 = 1024^2; 0.5 = 512^2, 0.25 = 256^2] [max engines]
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    emax = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    w = workloads.cfg2(scale)
    w.nt = 1000
    wav = w.wavelet()
    c = w.c.astype(np.float32)
    for ty in (8,):  # (rows per workgroup of the single-step kernel; the fused kernel ignores it)
        os.environ["FWI_STREAM_TY"] = str(ty)
        for E in [e for e in (1, 2, 3, 4, 6) if e <= emax]:
            engines = [Engine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml) for _ in range(E)]
            for e in engines:
                e.set_model(c)
                e.forward(None, (w.src_idx, wav), w.rec_idx, save=False)

            def work(e, reps=3):
                for _ in range(reps):
                    e.forward(None, (w.src_idx, wav), w.rec_idx, save=False)

            t0 = time.perf_counter()
            th = [threading.Thread(target=work, args=(e,)) for e in engines]
            [t.start() for t in th]
            [t.join() for t in th]
            el = time.perf_counter() - t0
            shots = 3 * E
            print("TY=%2d  %d engine(s): %.2f ms per shot aggregate (%.2f us/step/shot), %.1f Gpts/s aggregate" % (
                ty, E, 1e3 * el / shots, 1e6 * el / shots / w.nt, shots * w.updates_per_shot / el / 1e9), flush=True)
            for e in engines:
                e.close()


if __name__ == "__main__":
    main()
