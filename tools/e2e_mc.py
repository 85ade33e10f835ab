import time, numpy as np, sys
sys.path.insert(0, '.')
from full_waveform_inversion_amd import source_inversion as si
rng = np.random.default_rng(0)
k, n, t, N = 21, 9, 512, 1 << 20
G = rng.standard_normal((k, n, t)); Ms = rng.standard_normal((n, N))
d = np.einsum("kjt,j->kt", G, Ms[:, 5])
si.score_samples(d, G, Ms[:, :1024], "VR", False, False)
for _ in range(4):
    t0 = time.perf_counter(); r = si.score_samples(d, G, Ms, "VR", False, False, return_timing=True); w = time.perf_counter() - t0
    print("wall %.1f ms kernel %.2f ms -> %.1f M samples/s e2e" % (w * 1e3, r[3], N / w / 1e6))
