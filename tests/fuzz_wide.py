"""Wide seeded fuzz against the C oracle (a script, not collected by pytest): `python tests/fuzz_wide.py SEED COUNT`
on a GPU box.  Random 2-D / 3-D grids incl. nx around 250-300, all orders, sponge widths, fp32 / fp64, plain /
checkpointed / strided imaging; forward + adjoint + gradient.  Round 1: seeds 1, 2, 3 x 600 and 777 x 300 pass."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from full_waveform_inversion_amd import Engine
from oracle import fwi_oracle as fo
from oracle.c_oracle import CPropagator

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 777)
bad = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 250):
    nd = int(rng.integers(2, 4))
    order = int(rng.choice([2, 4, 8]))
    hi = 90 if nd == 2 else 41
    shape = tuple(int(rng.integers(3, hi)) for _ in range(nd))
    if rng.random() < 0.2:
        shape = shape[:-1] + (int(rng.integers(250, 300)),)
    npml = int(rng.integers(0, max(1, min(shape) // 2)))
    nt = int(rng.choice([7, 8, 12, 16, 21, 24]))
    dtype = "float64" if rng.random() < 0.25 else "float32"
    c = 1500.0 + 2000.0 * rng.random(shape)
    dt = 0.6 * fo.cfl_dt(c.max(), 6.0, nd, order)
    nsrc, nrec = int(rng.integers(1, 4)), int(rng.integers(1, 9))
    src = np.stack([rng.integers(0, s, nsrc) for s in shape], 1)
    rec = np.stack([rng.integers(0, s, nrec) for s in shape], 1)
    w = rng.standard_normal((nt, nsrc))
    r = rng.standard_normal((nt, nrec))
    mode = rng.choice(["plain", "ckpt", "stride"])
    ck = int(rng.choice([3, 4, 5, 8])) if mode == "ckpt" else 0
    st = int(rng.choice([2, 3, 4, 5])) if mode == "stride" else 1
    p = CPropagator(c, 6.0, dt, order, npml, image_stride=st)
    d = p.forward(src, w, rec); a = p.adjoint(r); g = p.gradient()
    with Engine(shape, 6.0, dt, nt, order=order, npml=npml, sigma_max=p.sigma_max, dtype=dtype, ckpt_interval=ck,
                image_stride=st) as e:
        dg = e.forward(c, (src, w), rec, save=True); ag = e.adjoint(r); gg = e.gradient(); kern = e.kernel_name
    tol = 1e-11 if dtype == "float64" else 3e-5
    # quantities that the short run leaves below ~1e-20 (receivers the wave has not reached: 1e-57 precursors
    # of the wide stencil) are below fp32's range; they are not compared
    errs = [np.linalg.norm(x - y) / np.linalg.norm(y) if np.abs(y).max() > 1e-18 else 0.0
            for x, y in ((dg, d), (ag, a), (gg, g))]
    if max(errs) > tol or not np.all(np.isfinite(gg)):
        bad += 1
        print("FAIL", case, shape, order, npml, nt, dtype, mode, ck, st, kern, errs, flush=True)
print("done, failures:", bad)
