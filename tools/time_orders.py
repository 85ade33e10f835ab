#!/usr/bin/env python3
"""3-D 256^3 forward step time for the three spatial orders (stream kernel), one GPU."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from full_waveform_inversion_amd import Engine, cfl_dt, ricker  # noqa: E402

n, nt = 256, 300
c = np.full((n, n, n), 2000.0, np.float32)
for order in (2, 4, 8):
    dt = 0.8 * cfl_dt(2000.0, 10.0, 3, order)
    with Engine((n, n, n), 10.0, dt, nt, order=order) as e:
        e.set_model(c)
        w = ricker(nt, dt, 10.0)
        ms = []
        for _ in range(3):
            e.forward(None, ([[128, 128, 128]], w), [[8, 128, 128]], save=False)
            ms.append(e.last_loop_ms())
        us = 1e3 * float(np.median(ms[1:])) / nt
        print("O(%d) %s: %.2f us/step  %.1f Gpts/s  %.0f GB/s algorithmic (%.1f %% of 8 TB/s)" % (
            order, e.kernel_name, us, n ** 3 / us / 1e3, 16 * n ** 3 / us / 1e3, 16 * n ** 3 / us / 1e3 / 80))
