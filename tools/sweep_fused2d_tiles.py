import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from full_waveform_inversion_amd import Engine, workloads
for n in (256, 384, 512, 640, 768, 1024):
    for ft in ("auto", "64", "32", "16"):
        if ft == "auto": os.environ.pop("FWI_FUSED2D_TILE", None)
        else: os.environ["FWI_FUSED2D_TILE"] = ft
        w = workloads.cfg2(n / 1024.0); w.nt = 400
        wav = w.wavelet()
        with Engine(w.shape, w.h, w.dt, w.nt, order=8, npml=w.npml) as e:
            e.set_model(w.c.astype(np.float32))
            ms = []
            for r in range(3):
                e.forward(None, (w.src_idx[:1], wav), w.rec_idx, save=False); ms.append(e.last_loop_ms())
            us = 1e3 * min(ms) / w.nt
            print("n=%4d tile=%-4s %6.2f us/step %7.1f Gpts/s" % (n, ft, us, n * n / us / 1e3), flush=True)
