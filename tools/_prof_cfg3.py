import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from full_waveform_inversion_amd import Engine, shots as sh, workloads
w = workloads.cfg3(1.0, nshots=6)
wav = w.wavelet()
shots = [sh.Shot(w.src_idx[i:i+1], wav, w.rec_idx) for i in range(6)]
e = Engine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml)
sh.model_data(e, w.c.astype(np.float32), shots)
m0 = w.c_init.astype(np.float32)
log = []
of, oa = e.forward, e.adjoint
def f(*a, **k):
    t=time.perf_counter(); r=of(*a, **k); log.append(("fwd", 1e3*(time.perf_counter()-t), e.last_loop_ms())); return r
def a_(*a, **k):
    t=time.perf_counter(); r=oa(*a, **k); log.append(("adj", 1e3*(time.perf_counter()-t), e.last_loop_ms())); return r
e.forward, e.adjoint = f, a_
for trial in range(2):
    log.clear()
    t=time.perf_counter(); J, g = sh.misfit_and_gradient(e, m0, shots); tot=1e3*(time.perf_counter()-t)
    print("total %.1f ms" % tot, " ".join("%s %.1f(%.1f)" % x for x in log))
