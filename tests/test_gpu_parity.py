"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle.

"Oracle" = this build's own NumPy / C restatement (oracle/): the reference has
no wave-propagation path, so these tests say "vs the build's
oracle", never "vs the reference".  Tolerances: fp64 kernels 1e-10 (same
arithmetic, different summation order), fp32 kernels 1e-5 relative L2 on
seismograms, adjoint traces and gradient -- the figure BASELINE.json states.
"""
import os

import numpy as np
import pytest

from full_waveform_inversion_amd import Engine, FwiError, workloads
from oracle import fwi_oracle as fo
from oracle.c_oracle import CPropagator

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TOL32 = 1e-5   # BASELINE.json north_star: "within 1e-5 relative L2" -- FLAT on seismograms, F^T r and the gradient of a
               # shared residual since round 4 (a 3x slack on the last two hid a 3-5x regression: profiles/r03_parity.json
               # measures <= 5.9e-6 on every one of them)
TOL64 = 1e-10
FLOOR = 1e-30  # absolute floor of the fuzz comparisons: below fp32's range the oracle's fp64 numbers have no fp32 counterpart


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) /
                 np.linalg.norm(np.asarray(b, np.float64)))


def poison_device_memory(mb=512):
    """Fill a chunk of free device memory with NaN bit patterns and release it: the next context's hipMalloc is
    likely to get those pages, so a kernel that reads a buffer (or a halo) nobody initialised yields NaN instead of
    the zeros a fresh process happens to see.  (How the option fuzz found the increment-form checkpoint buffer.)"""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    ptr = C.c_void_p()
    n = C.c_size_t(mb << 20)
    if hip.hipMalloc(C.byref(ptr), n) == 0:
        hip.hipMemset(ptr, 0xFF, n)
        hip.hipDeviceSynchronize()
        hip.hipFree(ptr)


def run_gpu(c, h, dt, order, npml, sigma_max, src, w, rec, residual=None, dtype="float32",
            kernel="auto", zchunk=0):
    nt = w.shape[0]
    with Engine(c.shape, h, dt, nt, order=order, npml=npml, sigma_max=sigma_max, dtype=dtype,
                kernel=kernel, zchunk=zchunk) as e:
        d = e.forward(c, (src, w), rec, save=residual is not None)
        out = {"seis": d, "kernel": e.kernel_name}
        if residual is not None:
            out["adj_src"] = e.adjoint(residual)
            out["grad_c"] = e.gradient("velocity")
            out["grad_m"] = e.gradient("slowness2")
    return out


# ---------------------------------------------------------------------------
# committed golden fixtures (inputs + oracle outputs travel to the GPU box)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["g2d_o8", "g2d_o2", "g3d_o8", "g3d_o4"])
@pytest.mark.parametrize("dtype,kernel,tol", [("float64", "point", TOL64), ("float32", "point", TOL32),
                                              ("float32", "auto", TOL32)])
def test_golden(gpu, name, dtype, kernel, tol):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    o = run_gpu(z["c"], float(z["h"]), float(z["dt"]), int(z["order"]), int(z["npml"]),
                float(z["sigma_max"]), z["src_idx"], z["wavelet"], z["rec_idx"], z["residual"],
                dtype=dtype, kernel=kernel)
    assert rel(o["seis"], z["seis"]) < tol
    assert rel(o["adj_src"], z["adj_src"]) < tol
    assert rel(o["grad_c"], z["grad_c"]) < tol
    assert rel(o["grad_m"], z["grad_m"]) < tol


# ---------------------------------------------------------------------------
# seeded random cases against the oracle run on the spot, incl. ragged shapes
# ---------------------------------------------------------------------------
SHAPES = [
    ((33, 47), 8, 6), ((64, 64), 2, 0), ((50, 30), 4, 5),          # 2-D
    ((20, 17, 23), 8, 4), ((16, 16, 64), 8, 0), ((12, 40, 36), 2, 3),  # 3-D, ragged / nx%4 != 0
    ((24, 24, 260), 8, 5),                                          # more than one 256-wide x tile
    ((9, 8, 8), 8, 0),                                              # grid barely wider than the stencil
]


@pytest.mark.parametrize("shape,order,npml", SHAPES)
@pytest.mark.parametrize("dtype,tol", [("float64", TOL64), ("float32", TOL32)])
def test_random_model_vs_oracle(gpu, shape, order, npml, dtype, tol):
    rng = np.random.default_rng(hash((shape, order)) % 2 ** 32)
    nd = len(shape)
    c = 1500.0 + 1500.0 * rng.random(shape)
    h = 7.5
    dt = 0.7 * fo.cfl_dt(c.max(), h, nd, order)
    nt = 70
    src = np.stack([rng.integers(0, s, 3) for s in shape], 1)
    src[2] = src[1]  # two sources on the same node: amplitudes add
    rec = np.stack([rng.integers(0, s, 11) for s in shape], 1)
    rec[0] = src[0]
    w = rng.standard_normal((nt, 3)) * fo.ricker(nt, dt, 20.0)[:, None]
    p = CPropagator(c, h, dt, order, npml)
    d = p.forward(src, w, rec)
    r = d + 0.1 * np.abs(d).max() * rng.standard_normal(d.shape)
    a = p.adjoint(r)
    o = run_gpu(c, h, dt, order, npml, p.sigma_max, src, w, rec, r, dtype=dtype)
    assert rel(o["seis"], d) < tol
    assert rel(o["adj_src"], a) < tol
    assert rel(o["grad_c"], p.gradient()) < tol


# The 8-row tile of the 3-D stream kernel (FWI_STREAM_TY=8) is what stream_default_tuning picks for every grid past the
# Infinity Cache (>= 272^3) -- sizes no oracle finishes in seconds, so every oracle comparison above ran 4-row tiles and
# the 8-row shape met the oracle only through the translation property (the kernel against itself).  The creation-time
# hook brings that tile shape to the oracle-sized grids: same bodies, same bar (VERDICT r03 missing #1).
@pytest.mark.parametrize("shape,order,npml", [c for c in SHAPES if len(c[0]) == 3])
@pytest.mark.parametrize("dtype,tol", [("float64", TOL64), ("float32", TOL32)])
def test_random_model_vs_oracle_with_8_row_tiles(gpu, monkeypatch, shape, order, npml, dtype, tol):
    monkeypatch.setenv("FWI_STREAM_TY", "8")
    test_random_model_vs_oracle(gpu, shape, order, npml, dtype, tol)


@pytest.mark.parametrize("zchunk", [0, 5, 16, 1000])
@pytest.mark.parametrize("dtype,tol", [("float32", 2e-6), ("float64", 1e-12)])
def test_stream_kernel_matches_point_kernel(gpu, zchunk, dtype, tol):
    """Same inputs through both stencil kernels (and several z-chunkings), fp32 and fp64."""
    w = workloads.cfg4(0.25, npml=6)  # 64^3
    wav = w.wavelet(np.dtype(dtype).type)
    a = run_gpu(w.c, w.h, w.dt, w.order, w.npml, None, w.src_idx, wav, w.rec_idx, kernel="point", dtype=dtype)
    b = run_gpu(w.c, w.h, w.dt, w.order, w.npml, None, w.src_idx, wav, w.rec_idx, kernel="stream",
                zchunk=zchunk, dtype=dtype)
    assert a["kernel"] == "step_point" and b["kernel"] == "step3d_stream"
    assert rel(b["seis"], a["seis"]) < tol


@pytest.mark.parametrize("shape,dtype,kern,tol", [((18, 21, 37), "float32", "step3d_stream", TOL32),
                                                  ((18, 21, 37), "float64", "step3d_stream", TOL64),
                                                  ((14, 9, 261), "float32", "step3d_stream", TOL32),
                                                  ((40, 131), "float32", "step2d_", TOL32)])
def test_stream_kernels_take_any_nx(gpu, shape, dtype, kern, tol):
    """kernel="stream" on grids with nx % 4 != 0 (compact rows padded to a multiple of 4, zero pad columns that
    stay zero because C = 0 there): forward + adjoint + gradient vs the C oracle, 12 steps so that the 2-D
    case runs the 4-steps-per-launch kernel."""
    rng = np.random.default_rng(7)
    nd = len(shape)
    c = 1500.0 + 1500.0 * rng.random(shape)
    h, order, npml, nt = 7.5, 8, 4, 12
    dt = 0.7 * fo.cfl_dt(c.max(), h, nd, order)
    src = np.array([[s - 1 for s in shape], [s // 2 for s in shape]])   # one source in the last column
    rec = np.stack([rng.integers(0, s, 9) for s in shape], 1)
    rec[0] = src[0]
    w = rng.standard_normal((nt, 2))
    p = CPropagator(c, h, dt, order, npml)
    d = p.forward(src, w, rec)
    r = d + 0.1 * np.abs(d).max() * rng.standard_normal(d.shape)
    a = p.adjoint(r)
    o = run_gpu(c, h, dt, order, npml, p.sigma_max, src, w, rec, r, dtype=dtype, kernel="stream")
    assert o["kernel"].startswith(kern)
    assert rel(o["seis"], d) < tol and rel(o["adj_src"], a) < tol and rel(o["grad_c"], p.gradient()) < tol


# ---------------------------------------------------------------------------
# BASELINE.json's configs, scaled so the oracle finishes in seconds
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("maker,scale", [(workloads.cfg1, 1.0), (workloads.cfg2, 0.25),
                                         (workloads.cfg4, 0.375), (workloads.cfg5, 0.25)])
def test_baseline_configs_scaled(gpu, maker, scale):
    w = maker(scale)
    wav = w.wavelet(np.float64)
    src = w.src_idx[:1]
    p = CPropagator(w.c, w.h, w.dt, w.order, w.npml)
    d = p.forward(src, wav, w.rec_idx, save=False)
    c0 = w.c_init if w.c_init is not None else w.c * 1.02
    p0 = CPropagator(c0, w.h, w.dt, w.order, w.npml, sigma_max=p.sigma_max)
    d0 = p0.forward(src, wav, w.rec_idx)
    p0.adjoint(d0 - d)
    nt = w.nt
    with Engine(w.shape, w.h, w.dt, nt, order=w.order, npml=w.npml, sigma_max=p.sigma_max) as e:
        dg = e.forward(w.c, (src, wav), w.rec_idx, save=False)
        assert rel(dg, d) < TOL32
        dg0 = e.forward(c0, (src, wav), w.rec_idx, save=True)
        assert rel(dg0, d0) < TOL32
        e.adjoint((d0 - d).astype(np.float32))   # same residual for both paths
        assert rel(e.gradient(), p0.gradient()) < TOL32


# ---------------------------------------------------------------------------
# size-independent properties at (or near) BASELINE.json's full sizes
# ---------------------------------------------------------------------------
def test_full_size_3d_adjoint_identity_and_linearity(gpu):
    """256^3 O(8), 300 steps: <F s, r> = <s, F^T r> and F(2s) = 2 F(s), no oracle needed."""
    w = workloads.cfg4(1.0, npml=16)
    nt = 300
    rng = np.random.default_rng(0)
    wav = w.wavelet()[:nt]
    src = np.array([[40, 128, 128]])  # 32 cells below the receiver patch: arrivals by step ~170
    with Engine(w.shape, w.h, w.dt, nt, order=8, npml=16) as e:
        d = e.forward(w.c.astype(np.float32), (src, wav), w.rec_idx, save=False)
        assert e.kernel_name == "step3d_stream"
        assert np.abs(d).max() > 1e-12
        d2 = e.forward(None, (src, 2 * wav), w.rec_idx, save=False)
        assert rel(d2, 2 * d) < 1e-6
        r = (rng.standard_normal(d.shape) * np.abs(d).max()).astype(np.float32)
        a = e.adjoint(r, image=False)
        lhs = e.dot(d, r)
        rhs = e.dot(wav[:, None], a)
        assert abs(lhs - rhs) < 1e-4 * max(abs(lhs), abs(rhs))


def _band_limited_residual(d, seed):
    """A residual both paths are handed: the data's own spectrum and moveout with random trace weights."""
    rng = np.random.default_rng(seed)
    return np.asarray(d, np.float64) * rng.uniform(0.5, 1.5, size=(1, d.shape[1])) + 0.1 * np.roll(d, 3, axis=0)


def test_full_size_3d_headline_config_vs_c_oracle(gpu):
    """BASELINE configs[3] at FULL size (256^3, O(8), 1000 steps, what bench.py times): the OpenMP C
    oracle needs ~10 s for it, so the headline run itself is checked, not a scaled copy."""
    w = workloads.cfg4(1.0)
    wav = w.wavelet(np.float64)
    p = CPropagator(w.c, w.h, w.dt, w.order, w.npml)
    d = p.forward(w.src_idx, wav, w.rec_idx, save=False)
    with Engine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml) as e:
        dg = e.forward(w.c, (w.src_idx, wav), w.rec_idx, save=False)
        assert e.kernel_name == "step3d_stream"
    assert np.abs(d).max() > 0
    assert rel(dg, d) < TOL32


def test_full_size_3d_grid_adjoint_and_gradient_vs_c_oracle(gpu, monkeypatch):
    """The full 256^3 O(8) grid with the sponge, forward(save) + adjoint + gradient against the C oracle on
    the SAME residual -- with the tuned tile shape (4 rows at this size) and with the 8-row tiles of larger grids.  400 of the 1000 steps: the oracle's store of the forward term is nt x 128 MiB of host
    memory (51 GiB here); the source sits 32 cells under the receiver patch so that the data, the adjoint field
    and their correlation are all well developed within those steps."""
    w = workloads.cfg4(1.0, npml=16)
    nt = 400
    wav = w.wavelet(np.float64)[:nt]
    src = np.array([[40, 128, 128]])
    p = CPropagator(w.c, w.h, w.dt, w.order, w.npml)
    d = p.forward(src, wav, w.rec_idx, save=True)
    assert np.abs(d).max() > 0
    r = _band_limited_residual(d, 1)
    a = p.adjoint(r)
    g = p.gradient("velocity")
    p.q_store = None
    assert np.abs(g).max() > 0 and np.abs(a).max() > 0
    for stream_ty in (None, "8"):  # (one oracle run serves both tile shapes)
        if stream_ty:
            monkeypatch.setenv("FWI_STREAM_TY", stream_ty)
        with Engine(w.shape, w.h, w.dt, nt, order=w.order, npml=w.npml, sigma_max=p.sigma_max) as e:
            dg = e.forward(w.c, (src, wav), w.rec_idx, save=True)
            assert e.kernel_name == "step3d_stream"
            ag = e.adjoint(r)
            gg = e.gradient("velocity")
        assert rel(dg, d) < TOL32, (stream_ty, rel(dg, d))
        assert rel(ag, a) < TOL32, (stream_ty, rel(ag, a))
        assert rel(gg, g) < TOL32, (stream_ty, rel(gg, g))


def test_full_size_2d_vs_c_oracle(gpu):
    """cfg2 at FULL size (1024^2, O(8) + sponge, 2000 steps): seismograms, F^T r and the gradient of a
    shared residual against the C oracle (whose forward-term store is 16 GiB of host memory here)."""
    w = workloads.cfg2(1.0)
    nt = w.nt
    wav = w.wavelet(np.float64)[:nt]
    p = CPropagator(w.c, w.h, w.dt, w.order, w.npml)
    d = p.forward(w.src_idx, wav, w.rec_idx, save=True)
    r = _band_limited_residual(d, 2)
    a = p.adjoint(r)
    g = p.gradient("velocity")
    p.q_store = None
    o = run_gpu(w.c, w.h, w.dt, w.order, w.npml, p.sigma_max, w.src_idx, wav, w.rec_idx, residual=r)
    assert o["kernel"] == "step2d_fused"
    assert rel(o["seis"], d) < TOL32
    assert rel(o["adj_src"], a) < TOL32
    assert rel(o["grad_c"], g) < TOL32


@pytest.mark.parametrize("maker,scale,nshots", [(workloads.cfg3, 0.25, 2), (workloads.cfg5, 0.25, 4)])
def test_end_to_end_fp64_engine_flat_1e5(gpu, maker, scale, nshots):
    """north_star's bar taken literally: observed data are the shared INPUT, each path forms its OWN residual
    (so the forward error enters the adjoint source amplified by |d| / |r|), and misfit and gradient must
    still agree to a flat 1e-5.  The fp64 engine meets it with margin; for the fp32 engine see
    test_shot_loop_gradient_vs_oracle and profiles/r02_parity.json."""
    from full_waveform_inversion_amd import shots as sh
    w = maker(scale, nshots=nshots)
    wav = w.wavelet(np.float64)
    J_ref, g_ref, dobs = 0.0, 0.0, []
    for i in range(nshots):
        pt = CPropagator(w.c, w.h, w.dt, w.order, w.npml)
        dobs.append(pt.forward(w.src_idx[i:i + 1], wav, w.rec_idx, save=False))
        pi = CPropagator(w.c_init, w.h, w.dt, w.order, w.npml, sigma_max=pt.sigma_max)
        r = pi.forward(w.src_idx[i:i + 1], wav, w.rec_idx, save=True) - dobs[-1]
        pi.adjoint(r)
        J_ref += 0.5 * float(np.sum(r * r))
        g_ref = g_ref + pi.gradient("velocity")
    shots = [sh.Shot(w.src_idx[i:i + 1], wav, w.rec_idx, dobs[i]) for i in range(nshots)]
    with Engine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml, sigma_max=pt.sigma_max,
                dtype="float64") as e:
        J, g = sh.misfit_and_gradient(e, w.c_init, shots)
    assert abs(J - J_ref) < 1e-5 * J_ref
    assert rel(g, g_ref) < 1e-5


@pytest.mark.parametrize("maker,scale", [(workloads.cfg2, 0.5), (workloads.cfg2, 1.0), (workloads.cfg4, 0.5),
                                         (workloads.cfg5, 0.25), (workloads.cfg5, 0.5)])
def test_end_to_end_fp32_increment_form_flat_1e5(gpu, maker, scale):
    """north_star's bar on an fp32 PRODUCTION mode: update_form="increment".  Observed data are
    the shared input, each path forms its OWN residual, and seismograms, misfit and gradient agree with the oracle
    to a FLAT 1e-5 -- every BASELINE config family, configs[1] at full size.  What made this possible in round 3 is
    the normalisation of the star weights (fwi_api.hip base_args): the rounded a_k / h^2 were a systematic operator
    perturbation that dominated the fp32 error (profiles/r03_parity.json; e.g. 1024^2 x 2000: 2.0e-5 -> 4.5e-6)."""
    w = maker(scale)
    wav = w.wavelet(np.float64)
    src = w.src_idx[:1]
    c0 = w.c_init if w.c_init is not None else w.c * (1.0 + 0.03 * np.sin(np.indices(w.shape).sum(0) / 9.0))
    p = CPropagator(w.c, w.h, w.dt, w.order, w.npml)
    d_obs = p.forward(src, wav, w.rec_idx, save=False)
    p0 = CPropagator(c0, w.h, w.dt, w.order, w.npml, sigma_max=p.sigma_max)
    d0 = p0.forward(src, wav, w.rec_idx)
    r = d0 - d_obs
    p0.adjoint(r)
    g0, J0 = p0.gradient(), 0.5 * float(np.sum(r * r))
    with Engine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml, sigma_max=p.sigma_max,
                update_form="increment") as e:
        dg = e.forward(c0, (src, wav), w.rec_idx, save=True)
        rg = dg.astype(np.float64) - d_obs
        e.adjoint(rg)
        gg = e.gradient()
    assert rel(dg, d0) < 1e-5
    assert abs(0.5 * float(np.sum(rg * rg)) - J0) < 1e-5 * J0
    assert rel(gg, g0) < 1e-5, rel(gg, g0)


# ---------------------------------------------------------------------------
# edge cases and error behaviour
# ---------------------------------------------------------------------------
def test_empty_point_sets(gpu):
    c = np.full((16, 16, 16), 2000.0, np.float32)
    with Engine(c.shape, 10.0, 1e-3, 8) as e:
        d = e.forward(c, (np.zeros((0, 3), np.int32), np.zeros((8, 0), np.float32)), [[3, 3, 3]])
        assert d.shape == (8, 1) and not d.any()
        d = e.forward(None, ([[8, 8, 8]], np.ones(8, np.float32)), np.zeros((0, 3), np.int32))
        assert d.shape == (8, 0)


def test_gradient_accumulates_over_shots_and_resets(gpu):
    w = workloads.cfg2(0.125, nshots=3)
    wav = w.wavelet()
    with Engine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml) as e:
        e.set_model(w.c.astype(np.float32))
        gs = []
        for s in range(3):
            e.reset_gradient()
            d = e.forward(None, (w.src_idx[s:s + 1], wav), w.rec_idx)
            e.adjoint(d)
            gs.append(e.gradient())
        e.reset_gradient()
        for s in range(3):
            d = e.forward(None, (w.src_idx[s:s + 1], wav), w.rec_idx)
            e.adjoint(d)
        assert rel(e.gradient(), gs[0] + gs[1] + gs[2]) < 1e-5
        e.reset_gradient()
        assert not e.gradient().any()


def test_error_paths(gpu):
    c = np.full((16, 16), 2000.0, np.float32)
    with Engine(c.shape, 10.0, 1e-3, 8) as e:
        with pytest.raises(FwiError) as ei:
            e.adjoint(np.zeros((0, 0), np.float32))
        assert ei.value.code == 3  # FWI_ESTATE: nothing to adjoin
        with pytest.raises(FwiError) as ei:
            e.forward(c, ([[16, 0]], np.ones(8, np.float32)), [[1, 1]])
        assert ei.value.code == 1 and "outside the grid" in str(ei.value)
        with pytest.raises(FwiError) as ei:
            e.forward(c, ([[1, 1]], np.ones(9, np.float32)), [[1, 1]])
        assert ei.value.code == 1  # nt > nt_max
        e.forward(c, ([[8, 8]], np.ones(8, np.float32)), [[1, 1]], save=False)
        with pytest.raises(FwiError) as ei:
            e.adjoint(np.zeros((8, 1), np.float32), image=True)
        assert ei.value.code == 3  # imaging needs save=True
        with pytest.raises(FwiError):
            e.set_model(np.zeros_like(c))  # velocity must be > 0
    with pytest.raises(FwiError):  # the fp64 float4-style kernels exist in 3-D only
        Engine((8, 8), 10.0, 1e-3, 4, dtype="float64", kernel="stream")
    with Engine((8, 8, 9), 10.0, 1e-3, 4, dtype="float64", kernel="stream") as e:  # any nx (padded compact rows)
        assert e.kernel_name == "step3d_stream"


def test_dot_product_reduction(gpu):
    rng = np.random.default_rng(1)
    a = rng.standard_normal(1_000_003).astype(np.float32)
    b = rng.standard_normal(1_000_003).astype(np.float32)
    with Engine((8, 8), 10.0, 1e-3, 2) as e:
        e.set_model(np.full((8, 8), 2000.0, np.float32))
        got = e.dot(a, b)
    ref = float(np.dot(a.astype(np.float64), b.astype(np.float64)))
    assert abs(got - ref) < 1e-9 * np.sqrt(a.size) * 10 + 1e-12 * abs(ref)


def test_single_rank_rccl_allreduce(gpu):
    """RCCL communicator of one rank: all-reduce must leave the gradient unchanged."""
    w = workloads.cfg2(0.125)
    wav = w.wavelet()
    with Engine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml) as e:
        d = e.forward(w.c.astype(np.float32), (w.src_idx, wav), w.rec_idx)
        e.adjoint(d)
        g0 = e.gradient()
        with pytest.raises(FwiError) as ei:  # no communicator yet
            e.comm_info()
        assert ei.value.code == 3
        # the production exchange object over a one-rank control plane: unique id -> ncclCommInitRank ->
        # what RCCL itself reports (ncclCommCount / ncclCommUserRank), the figure bench.py echoes
        from full_waveform_inversion_amd.rendezvous import Rendezvous
        from full_waveform_inversion_amd.shots import RcclExchange
        ex = RcclExchange(e, Rendezvous(0, 1))
        assert ex.rccl_ranks == 1 and e.comm_info() == (1, 0)
        e.allreduce_gradient()
        assert np.array_equal(e.gradient(), g0)
        assert e.allreduce_f64([1.5, -2.0]) == [1.5, -2.0]
        assert e.allreduce_f64([1.5, -2.0], op="max") == [1.5, -2.0]
        assert ex.reduce_device(e, 3.25) == 3.25
        e.comm_abort()
        with pytest.raises(FwiError):  # aborted: the exchange is gone, not silently skipped
            e.allreduce_gradient()
        with pytest.raises(FwiError) as ei:  # a rank outside the communicator is refused before RCCL is entered
            e.comm_init(2, 2, Engine.comm_unique_id())
        assert ei.value.code == 1


# ---------------------------------------------------------------------------
# shot loop + optimiser on the real engine (BASELINE configs[2] / [4], scaled)
# ---------------------------------------------------------------------------
def test_shot_loop_gradient_vs_oracle(gpu):
    """cfg3 scaled: 4 shots, summed misfit and gradient at the smoothed start model.

    End to end (observed data is the shared INPUT, each path forms its own residual): the
    fp32 forward error (~1e-6 of |d|, see tests/parity_report.py) enters the residual
    amplified by |d| / |r|, so the bar is scaled by that measured ratio instead of the flat
    1e-5 that holds for the seismograms and for the gradient of a given residual
    (test_baseline_configs_scaled).
    """
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _oracle_engine import OracleEngine
    from full_waveform_inversion_amd import shots as sh

    w = workloads.cfg3(0.125, nshots=4)  # 128 x 128, 250 steps
    wav = w.wavelet(np.float64)
    shots = [sh.Shot(w.src_idx[i:i + 1], wav, w.rec_idx) for i in range(4)]
    o = OracleEngine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml)
    sh.model_data(o, w.c, shots)
    c0 = 0.97 * w.c_init  # start model 3 % slow: a residual comparable to the data
    J0, g0 = sh.misfit_and_gradient(o, c0, shots)
    amp = np.sqrt(sum(np.sum(s.d_obs ** 2) for s in shots) / (2 * J0))  # |d| / |r|
    with Engine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml, sigma_max=o.sigma_max) as e:
        J1, g1 = sh.misfit_and_gradient(e, c0, shots)
    assert amp < 30
    assert abs(J1 - J0) < 2 * TOL32 * amp * J0
    assert rel(g1, g0) < TOL32 * amp


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_device_residual_and_misfit_equal_the_host_ones(gpu, dtype):
    """fwi_misfit_l2: residual formed and reduced on the device (wave-shuffle reduction), back-propagated
    by adjoint(None) -- the same residual bit for bit as d_syn - d_obs on the host, hence the same gradient."""
    from full_waveform_inversion_amd import objectives as ob, shots as sh
    w = workloads.cfg5(0.125, nshots=4)
    wav = w.wavelet(np.dtype(dtype).type)
    shots = [sh.Shot(w.src_idx[i:i + 1], wav, w.rec_idx) for i in range(3)]
    with Engine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml, dtype=dtype) as e:
        sh.model_data(e, w.c, shots)
        J_dev, g_dev = sh.misfit_and_gradient(e, w.c_init, shots)                       # device residual
        J_host, g_host = sh.misfit_and_gradient(e, w.c_init, shots, objective=lambda s, d: ob.l2(s, d))
        assert abs(J_dev - J_host) <= 1e-12 * J_host and np.array_equal(g_dev, g_host)
        d = e.forward(None, (w.src_idx[:1], wav), w.rec_idx, save=False)
        assert abs(e.misfit_l2(np.zeros_like(d)) - 0.5 * float(np.sum(d.astype(np.float64) ** 2))) <= 1e-12 * np.sum(d ** 2)
        e.adjoint(None, image=False)                # consumes the device residual
        with pytest.raises(FwiError) as ei:         # ... so a second adjoint(None) has nothing to propagate
            e.adjoint(None, image=False)
        assert ei.value.code == 1
        e.forward(None, (w.src_idx[:1], wav), w.rec_idx, save=False)
        with pytest.raises(FwiError):               # a new forward invalidates the old residual too
            e.adjoint(None, image=False)


@pytest.mark.parametrize("shape,order,npml,dtype,tol,kern", [
    ((40, 36, 256), 8, 6, "float32", TOL32, "step3d_stream"),   # full 256-column tiles
    ((33, 29, 50), 8, 5, "float32", TOL32, "step3d_stream"),    # ragged tiles
    ((30, 28, 36), 4, 0, "float32", TOL32, "step3d_stream"),    # no damping
    ((30, 28, 36), 8, 4, "float64", TOL64, "step_point"),
    ((70, 90), 8, 8, "float32", TOL32, "step2d_fused"),         # 2-D fp32: the fused kernel carries this form too
    ((70, 91), 4, 6, "float32", TOL32, "step2d_fused"),
    ((70, 90), 2, 0, "float64", TOL64, "step_point")])
def test_increment_form_matches_the_oracle(gpu, shape, order, npml, dtype, tol, kern):
    """fwi_config.update_form = INCREMENT: the recursion carried as (u, v = u - u_prev).  Same mathematics as
    the oracle's standard form, so the same tolerances hold for seismograms, F^T r and the gradient -- with
    checkpointing (the snapshots hold (u, v)) as without."""
    rng = np.random.default_rng(5)
    c = 2000.0 + 600.0 * rng.random(shape)
    h = 10.0
    dt = 0.7 * fo.cfl_dt(c.max(), h, len(shape), order)
    nt = 60
    src = np.array([[s // 2 for s in shape]])
    rec = np.array([[max(npml, 2) + 1] + [s // 3 for s in shape[1:]], [s // 2 + 2 for s in shape]])
    wav = fo.ricker(nt, dt, 0.12 / dt / 8)
    p = fo.Propagator(c, h, dt, order, npml)
    d = p.forward(src, wav, rec)
    r = d * rng.uniform(0.5, 1.5, size=(1, 2))
    a = p.adjoint(r)
    g = p.gradient()
    for K in (0, 16):
        with Engine(shape, h, dt, nt, order=order, npml=npml, sigma_max=p.sigma_max, dtype=dtype,
                    update_form="increment", ckpt_interval=K) as e:
            dg = e.forward(c, (src, wav), rec, save=True)
            assert e.kernel_name == kern
            ag = e.adjoint(r)
            gg = e.gradient()
        assert rel(dg, d) < tol and rel(ag, a) < tol and rel(gg, g) < tol, (K, rel(dg, d), rel(ag, a), rel(gg, g))
    if len(shape) == 2:
        with pytest.raises(FwiError):  # the 2-D stream kernels have no increment form: refused, not ignored
            Engine(shape, h, dt, nt, order=order, npml=npml, sigma_max=p.sigma_max, dtype=dtype,
                   update_form="increment", kernel="stream")


@pytest.mark.parametrize("shape,order,npml", [((40, 36, 256), 8, 6), ((33, 29, 50), 8, 5), ((30, 28, 36), 4, 0)])
def test_increment_form_matches_the_oracle_with_8_row_tiles(gpu, monkeypatch, shape, order, npml):
    monkeypatch.setenv("FWI_STREAM_TY", "8")
    test_increment_form_matches_the_oracle(gpu, shape, order, npml, "float32", TOL32, "step3d_stream")


@pytest.mark.parametrize("shape,order,nt,zc", [
    ((40, 36, 256), 8, 40, 0),     # one full-row tile per y strip
    ((40, 36, 256), 8, 41, 16),    # odd step count (last step single), z chunks of 16: sources on chunk seams
    ((33, 29, 50), 8, 30, 7),      # ragged everything, chunks shorter than the halo stack
    ((24, 21, 300), 8, 26, 0),     # rows wider than 256 columns: x tiles with their own halo
    ((30, 28, 36), 4, 30, 0),
    ((30, 28, 36), 2, 30, 0)])
def test_two_steps_per_pass_3d_matches_the_oracle(gpu, monkeypatch, shape, order, nt, zc):
    """3-D temporal blocking (fwi_pair3d.hip): forward sweeps that keep nothing for imaging advance two time
    steps per pass.  Same arithmetic as the single-step kernel, so the same tolerance -- with sources on tile and
    chunk seams (every neighbouring workgroup must see the first step's source before its second step) and
    receivers anywhere."""
    monkeypatch.setenv("FWI_STREAM_PAIR", "1")
    if zc:
        monkeypatch.setenv("FWI_PAIR_ZCHUNK", str(zc))
    rng = np.random.default_rng(9)
    c = 2000.0 + 600.0 * rng.random(shape)
    h = 10.0
    dt = 0.7 * fo.cfl_dt(c.max(), h, 3, order)
    nz, ny, nx = shape
    src = np.array([[nz // 2, ny // 2, nx // 2], [15, 7, 3], [16, 8, nx - 2], [min(nz - 1, 17), 15, nx // 2 + 1],
                    [0, 0, 0], [nz - 1, ny - 1, nx - 1], [nz // 2, ny // 2, nx // 2]])  # (a duplicate node too)
    rec = np.stack([rng.integers(0, s, 12) for s in shape], 1)
    wav = np.stack([fo.ricker(nt, dt, 0.12 / dt / 8) * (1 + 0.2 * k) for k in range(len(src))], 1)
    d = fo.Propagator(c, h, dt, order, 0).forward(src, wav, rec, save=False)
    with Engine(shape, h, dt, nt, order=order, npml=0) as e:
        dg = e.forward(c, (src, wav), rec, save=False)
        dg2 = e.forward(None, (src, wav), rec, save=True)   # the single-step path of the same context
    assert rel(dg, d) < TOL32 and rel(dg2, d) < TOL32
    assert rel(dg, dg2) < 2e-6


@pytest.mark.parametrize("shape,npml,stride", [((40, 36, 256), 6, 1), ((33, 29, 50), 5, 1), ((33, 29, 50), 0, 3)])
def test_bf16_forward_term_store(gpu, shape, npml, stride):
    """fwi_config.store_dtype = BF16 (SURVEY s.8f-3 "snapshot compression"): the forward term is kept in bf16 --
    half the store, half its traffic -- and the source's own share of it is paired in closed form.  Against the
    oracle's restatement of exactly that (1e-4: a bf16 rounding that flips between the fp32 and the fp64 path is a
    0.4 % change of one sample) and against the exact gradient (the price of the compression: a few 1e-3)."""
    rng = np.random.default_rng(13)
    c = 2000.0 + 600.0 * rng.random(shape)
    h, order, nt = 10.0, 8, 60
    dt = 0.7 * fo.cfl_dt(c.max(), h, 3, order)
    src = np.array([[s // 2 for s in shape], [s // 3 for s in shape], [s // 2 for s in shape]])  # a duplicate node
    rec = np.stack([rng.integers(0, s, 6) for s in shape], 1)
    wav = np.stack([fo.ricker(nt, dt, 0.12 / dt / 8) * a for a in (1.0, 0.7, -0.4)], 1)
    pq = fo.Propagator(c, h, dt, order, npml, store_dtype="bf16", image_stride=stride)
    px = fo.Propagator(c, h, dt, order, npml, sigma_max=pq.sigma_max, image_stride=stride)
    d = pq.forward(src, wav, rec)
    r = d * rng.uniform(0.5, 1.5, size=(1, len(rec)))
    pq.adjoint(r)
    px.forward(src, wav, rec)
    px.adjoint(r)
    with Engine(shape, h, dt, nt, order=order, npml=npml, sigma_max=pq.sigma_max, store_dtype="bf16",
                image_stride=stride) as e:
        dg = e.forward(c, (src, wav), rec, save=True)
        assert e.kernel_name == "step3d_stream"
        ag = e.adjoint(r)
        gg = e.gradient()
    assert rel(dg, d) < TOL32
    assert rel(gg, pq.gradient()) < 1e-4
    assert 1e-5 < rel(gg, px.gradient()) < 1e-2
    for bad in (dict(dtype="float64"), dict(ckpt_interval=8), dict(update_form="increment"), dict(kernel="point"),
                dict(abc="cpml") if npml else dict(dtype="float64")):
        with pytest.raises(FwiError):
            Engine(shape, h, dt, nt, order=order, npml=npml, sigma_max=pq.sigma_max, store_dtype="bf16", **bad)
    with pytest.raises(FwiError):
        Engine((40, 40), h, dt, nt, store_dtype="bf16")
    with pytest.raises(FwiError):  # (round 4: the bf16 store exists for the O(8) stencil only)
        Engine(shape, h, dt, nt, order=4, npml=npml, sigma_max=pq.sigma_max, store_dtype="bf16")


@pytest.mark.parametrize("shape,npml,stride", [((40, 36, 256), 6, 1), ((33, 29, 50), 5, 1), ((33, 29, 50), 0, 3)])
def test_bf16_forward_term_store_with_8_row_tiles(gpu, monkeypatch, shape, npml, stride):
    monkeypatch.setenv("FWI_STREAM_TY", "8")
    test_bf16_forward_term_store(gpu, shape, npml, stride)


@pytest.mark.parametrize("shape", [(48, 40, 64), (96, 100)])
def test_duplicate_injection_nodes_are_bit_reproducible(gpu, shape):
    """Four sources on ONE node (and receivers doubling as adjoint sources on one node): every kernel sums a node's
    entries from one thread in a host-fixed order and issues one add per node, so repeated runs agree bit for bit --
    forward seismograms, adjoint source series and gradient."""
    rng = np.random.default_rng(21)
    nd = len(shape)
    c = (2000.0 + 500.0 * rng.random(shape)).astype(np.float32)
    h, order, nt = 10.0, 8, 64
    dt = 0.7 * fo.cfl_dt(float(c.max()), h, nd, order)
    node = [s // 2 for s in shape]
    src = np.array([node, node, [s // 3 for s in shape], node, node])
    rec = np.array([[s // 4 for s in shape]] * 3 + [[s // 2 + 3 for s in shape]])
    w = rng.standard_normal((nt, len(src))).astype(np.float32)
    r = rng.standard_normal((nt, len(rec))).astype(np.float32)
    runs = []
    for _ in range(3):
        with Engine(shape, h, dt, nt, order=order, npml=5) as e:
            d = e.forward(c, (src, w), rec, save=True)
            a = e.adjoint(r)
            runs.append((d, a, e.gradient()))
    for d, a, g in runs[1:]:
        assert np.array_equal(d, runs[0][0]) and np.array_equal(a, runs[0][1]) and np.array_equal(g, runs[0][2])


def test_lbfgs_inversion_on_gpu_reduces_misfit(gpu):
    """cfg5 scaled (3-D, smooth random model): 3 L-BFGS iterations with the GPU dot product."""
    from full_waveform_inversion_amd import shots as sh
    from full_waveform_inversion_amd.lbfgs import lbfgs

    w = workloads.cfg5(0.1875, nshots=4)  # 48^3
    wav = w.wavelet()
    shots = [sh.Shot(w.src_idx[i:i + 1], wav, w.rec_idx) for i in range(4)]
    with Engine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml) as e:
        sh.model_data(e, w.c.astype(np.float32), shots)
        fg = lambda m: sh.misfit_and_gradient(e, m, shots)  # noqa: E731
        _, _, log = lbfgs(fg, w.c_init.astype(np.float32), maxiter=3, history=3, first_step=40.0,
                          bounds=(1000.0, 5000.0), dot=e.dot)
    assert log[-1]["f"] < 0.6 * log[0]["f"], log


def test_2d_kernels_agree(gpu, monkeypatch):
    """2-D: one-thread-per-point, LDS-tiled float4 and 4-steps-per-launch fused kernels on the same
    inputs: ragged rows, two x tiles, sources near a corner and near tile seams, damping on."""
    rng = np.random.default_rng(5)
    shape = (133, 260)
    c = 1500.0 + 1500.0 * rng.random(shape)
    dt = 0.7 * fo.cfl_dt(c.max(), 5.0, 2, 8)
    nt = 120
    src = np.array([[60, 130], [3, 3], [63, 64]])
    rec = np.stack([rng.integers(0, s, 60) for s in shape], 1)
    rec[:3] = [[63, 63], [64, 64], [132, 259]]
    wav = np.stack([fo.ricker(nt, dt, 25.0), fo.ricker(nt, dt, 15.0), -fo.ricker(nt, dt, 20.0)], 1)
    res = rng.standard_normal((nt, 60))
    out = {}
    for name, kern, nofuse in (("point", "point", "1"), ("tile", "stream", "1"), ("fused", "stream", None)):
        if nofuse:
            monkeypatch.setenv("FWI_NO_FUSED2D", nofuse)
        else:
            monkeypatch.delenv("FWI_NO_FUSED2D", raising=False)
        out[name] = run_gpu(c, 5.0, dt, 8, 12, 900.0, src, wav, rec, residual=res, kernel=kern)
    assert [out[k]["kernel"] for k in ("point", "tile", "fused")] == ["step_point", "step2d_tile", "step2d_fused"]
    for k in ("tile", "fused"):
        assert rel(out[k]["seis"], out["point"]["seis"]) < 2e-6
        assert rel(out[k]["adj_src"], out["point"]["adj_src"]) < 2e-6
        assert rel(out[k]["grad_c"], out["point"]["grad_c"]) < 1e-5


def test_2d_fused_and_tile_kernels_are_bit_identical_with_interior_tiles(gpu, monkeypatch):
    """A 5 x 5-tile grid: the fused kernel's interior tiles (no damping anywhere in their extended region) take the
    plain update, the border tiles the damped one; both must give the bits of `step2d_tile` -- step counts off the
    multiple of 4 and checkpointed runs mix the two kernels within one shot."""
    rng = np.random.default_rng(8)
    shape = (300, 290)
    c = (1500.0 + 1500.0 * rng.random(shape)).astype(np.float32)
    dt = 0.7 * fo.cfl_dt(float(c.max()), 5.0, 2, 8)
    nt = 64
    src = np.array([[150, 140], [20, 30], [128, 192]])
    rec = np.stack([rng.integers(0, s, 40) for s in shape], 1)
    wav = rng.standard_normal((nt, 3)).astype(np.float32)
    out = {}
    monkeypatch.setenv("FWI_FUSED2D_SKIPD", "1")  # (by itself the variant is taken from 257 tiles on)
    for name, nofuse in (("tile", "1"), ("fused", None)):
        if nofuse:
            monkeypatch.setenv("FWI_NO_FUSED2D", nofuse)
        else:
            monkeypatch.delenv("FWI_NO_FUSED2D", raising=False)
        for form in ("standard", "increment"):
            with Engine(shape, 5.0, dt, nt, order=8, npml=10, sigma_max=900.0, update_form=form,
                        kernel=("auto" if form == "increment" else "stream")) as e:
                d = e.forward(c, (src, wav), rec, save=False)
                out[name, form] = (d, e.kernel_name)
    assert out["tile", "standard"][1] == "step2d_tile" and out["fused", "standard"][1] == "step2d_fused"
    assert np.array_equal(out["fused", "standard"][0], out["tile", "standard"][0])
    assert rel(out["fused", "increment"][0], out["tile", "increment"][0]) < 2e-6  # (point kernel: another summation order)


@pytest.mark.parametrize("shape,K", [((40, 36, 44), 7), ((40, 36, 44), 1), ((40, 36, 44), 500), ((96, 100), 16),
                                     ((96, 100), 7), ((96, 100), 8)])
def test_checkpointed_gradient_equals_store_all(gpu, shape, K):
    """SURVEY s.8f-3: snapshots every K steps + recomputation give the store-all gradient
    (K not dividing nt, K = 1, K > nt; 3-D stream kernel and 2-D tile kernel)."""
    rng = np.random.default_rng(11)
    nd = len(shape)
    c = 1800.0 + 1200.0 * rng.random(shape)
    dt = 0.7 * fo.cfl_dt(c.max(), 8.0, nd, 8)
    nt = 92  # a multiple of 4: 2-D contexts take the fused path when K is one too (16, 8), not for K = 7
    src = np.array([[s // 2 for s in shape], [s // 3 for s in shape]])
    rec = np.stack([rng.integers(0, s, 17) for s in shape], 1)
    wav = np.stack([fo.ricker(nt, dt, 22.0), -0.5 * fo.ricker(nt, dt, 16.0)], 1).astype(np.float32)
    out = []
    for ck in (0, K):
        with Engine(shape, 8.0, dt, nt, order=8, npml=5, sigma_max=700.0, ckpt_interval=ck) as e:
            d = e.forward(c, (src, wav), rec, save=True)
            r = (d * rng.standard_normal(1)).astype(np.float32) if not out else out[0][2]
            a = e.adjoint(r)
            out.append((d, a, r, e.gradient()))
    assert np.array_equal(out[0][0], out[1][0])
    assert rel(out[1][1], out[0][1]) < 1e-6
    assert rel(out[1][3], out[0][3]) < 1e-6


@pytest.mark.parametrize("shape", [(20, 17, 24), (20, 17, 23), (21, 30), (9, 5, 1)])
def test_device_vector_algebra(gpu, shape):
    """nx % 4 != 0: the device keeps model-shaped arrays with rows padded to a multiple of 4 (zeros in the
    pad); uploads, downloads, reductions and the clamp must not see the pad."""
    rng = np.random.default_rng(2)
    a, b = rng.standard_normal(shape).astype(np.float32), rng.standard_normal(shape).astype(np.float32)
    with Engine(shape, 10.0, 1e-3, 4) as e:
        e.vec_create(3)
        e.vec_upload(0, a)
        e.vec_upload(1, b)
        assert abs(e.vec_dot(0, 1) - float(np.sum(a.astype(np.float64) * b))) < 1e-9 * a.size
        assert e.vec_absmax(1) == float(np.abs(b).max())
        e.vec_axpby(1, 2.5, 0, -0.5)  # b = 2.5 a - 0.5 b
        assert np.allclose(e.vec_download(1), 2.5 * a - 0.5 * b, rtol=1e-6, atol=1e-6)
        e.vec_copy(2, 0)
        e.vec_clip(2, 0.25, 0.5)   # a clamp away from zero: pad columns must stay out of the reductions
        assert np.array_equal(e.vec_download(2), np.clip(a, 0.25, 0.5))
        clipped = np.clip(a, 0.25, 0.5).astype(np.float64)
        assert abs(e.vec_dot(2, 2) - float(np.sum(clipped * clipped))) < 1e-9 * a.size
        with pytest.raises(FwiError):
            e.vec_dot(0, 3)
        # model / gradient hand-over without leaving the device
        c = (2000.0 + 100.0 * a).astype(np.float32)
        e.vec_upload(0, c)
        e.set_model_vec(0)
        wav = fo.ricker(4, 1e-3, 30.0).astype(np.float32)
        pt = [[s // 2 for s in shape]]
        d = e.forward(None, (pt, wav), [[s // 3 for s in shape]])
        e.adjoint(np.ones_like(d))
        e.gradient_vec(1)
        g = e.gradient()
        assert np.all(np.isfinite(g)) and np.array_equal(e.vec_download(1), g)
        assert abs(e.vec_dot(1, 1) - float(np.sum(g.astype(np.float64) ** 2))) <= 1e-6 * float(np.sum(g.astype(np.float64) ** 2))


@pytest.mark.parametrize("shape,dtype,tol,nt", [((24, 20, 32), "float32", TOL32, 41), ((24, 20, 31), "float64", TOL64, 41),
                                                ((48, 64), "float32", TOL32, 41),    # 2-D, 40 steps fused + 1 single
                                                ((70, 131), "float32", TOL32, 40)])  # 2-D, 4 steps per launch
@pytest.mark.parametrize("stride", [2, 3, 7])
def test_image_stride_matches_the_oracle_definition(gpu, shape, dtype, tol, nt, stride):
    """fwi_config.image_stride: the forward term is stored / correlated every S-th step with weight S -- same
    gradient as the oracle with the same stride, for nt not a multiple of S, two shots accumulated."""
    rng = np.random.default_rng(11)
    nd = len(shape)
    c = 1800.0 + 900.0 * rng.random(shape)
    h, order, npml = 8.0, 8, 4
    dt = 0.6 * fo.cfl_dt(c.max(), h, nd, order)
    rec = np.stack([rng.integers(0, s, 7) for s in shape], 1)
    p = CPropagator(c, h, dt, order, npml, image_stride=stride)
    g_ref = np.zeros(shape)
    with Engine(shape, h, dt, nt, order=order, npml=npml, sigma_max=p.sigma_max, dtype=dtype,
                image_stride=stride) as e:
        for shot in range(2):
            src = np.stack([rng.integers(0, s, 2) for s in shape], 1)
            w = rng.standard_normal((nt, 2))
            d = p.forward(src, w, rec)
            r = d + 0.2 * np.abs(d).max() * rng.standard_normal(d.shape)
            a = p.adjoint(r)
            g_ref += p.gradient()
            dg = e.forward(c, (src, w), rec, save=True)
            ag = e.adjoint(r)
            assert rel(dg, d) < tol and rel(ag, a) < tol
        assert rel(e.gradient(), g_ref) < tol
        assert e.kernel_name == ("step3d_stream" if nd == 3 else "step2d_fused")


def test_image_stride_shrinks_the_store(gpu):
    """A store-all run that does not fit is refused; the same run with a stride allocates 1/S of it."""
    shape, nt = (256, 256, 256), 6000   # 6000 x 64 MiB = 375 GiB > 288 GB
    with Engine(shape, 10.0, 1e-3, nt) as e:
        e.set_model(np.full(shape, 2000.0, np.float32))
        with pytest.raises(FwiError) as ei:
            e.forward(None, ([[128, 128, 128]], np.zeros(8, np.float32)), [[8, 8, 8]], save=True)
        assert ei.value.code == 4 and "image_stride" in str(ei.value)
    with Engine(shape, 10.0, 1e-3, nt, image_stride=8) as e:   # 750 slots = 47 GiB
        e.set_model(np.full(shape, 2000.0, np.float32))
        w = fo.ricker(16, 1e-3, 30.0).astype(np.float32)
        d = e.forward(None, ([[128, 128, 128]], w), [[120, 128, 128]], save=True)
        e.adjoint(np.ones_like(d))
        assert np.isfinite(e.gradient()).all()


@pytest.mark.parametrize("nt", [5, 41, 42, 43])
def test_2d_step_counts_off_the_fused_multiple(gpu, nt):
    """nt % 4 != 0 in 2-D: the bulk runs 4 steps per launch, the last 1-3 steps one per launch; the imaging
    pairings of the two conventions (in-launch vs lagged) must tile the time axis exactly once."""
    rng = np.random.default_rng(nt)
    shape, h, order, npml = (60, 90), 8.0, 8, 5
    c = 1800.0 + 900.0 * rng.random(shape)
    dt = 0.6 * fo.cfl_dt(c.max(), h, 2, order)
    src = np.stack([rng.integers(4, s - 4, 2) for s in shape], 1)
    rec = np.stack([rng.integers(0, s, 9) for s in shape], 1)
    rec[:2] = src                      # receivers on the source nodes: data from the first step on
    rec[2] = src[0] + [2, 3]
    w = rng.standard_normal((nt, 2))
    p = CPropagator(c, h, dt, order, npml)
    d = p.forward(src, w, rec)
    r = d + 0.2 * np.abs(d).max() * rng.standard_normal(d.shape)
    a = p.adjoint(r)
    o = run_gpu(c, h, dt, order, npml, p.sigma_max, src, w, rec, r)
    assert o["kernel"] == "step2d_fused"
    assert rel(o["seis"], d) < TOL32 and rel(o["adj_src"], a) < TOL32 and rel(o["grad_c"], p.gradient()) < TOL32


def test_device_lbfgs_matches_host_lbfgs(gpu):
    """Same iterates from the device-resident and the host L-BFGS on a small 3-D inversion."""
    from full_waveform_inversion_amd import shots as sh
    from full_waveform_inversion_amd.lbfgs import lbfgs, lbfgs_device

    w = workloads.cfg5(0.1875, nshots=3)  # 48^3
    wav = w.wavelet()
    shots = [sh.Shot(w.src_idx[i:i + 1], wav, w.rec_idx) for i in range(3)]
    x0 = w.c_init.astype(np.float32)
    with Engine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml) as e:
        sh.model_data(e, w.c.astype(np.float32), shots)
        xh, fh, logh = lbfgs(lambda m: sh.misfit_and_gradient(e, m, shots), x0, maxiter=3, history=3,
                             first_step=40.0, bounds=(1000.0, 5000.0), dot=e.dot)
        xd, fd, logd = lbfgs_device(e, lambda xs, gs: sh.misfit_and_gradient_device(e, xs, gs, shots), x0,
                                    maxiter=3, history=3, first_step=40.0, bounds=(1000.0, 5000.0))
    assert [r["evals"] for r in logd] == [r["evals"] for r in logh]
    assert abs(fd - fh) < 1e-3 * fh and fd < 0.6 * logd[0]["f"]
    assert rel(xd, xh) < 1e-5


def test_store_too_large_is_a_clean_error(gpu):
    """A forward-term store that cannot fit must fail with FWI_ENOMEM and a usable message."""
    n = 512
    with Engine((n, n, n), 10.0, 1e-3, 100000, npml=0) as e:  # 100000 x 512^3 x 4 B = 50 TB
        e.set_model(np.full((n, n, n), 2000.0, np.float32))
        with pytest.raises(FwiError) as ei:
            e.forward(None, ([[8, 8, 8]], np.ones(4, np.float32)), [[9, 9, 9]], save=True)
        assert ei.value.code == 4 and "ckpt_interval" in str(ei.value)
        d = e.forward(None, ([[8, 8, 8]], np.ones(4, np.float32)), [[9, 9, 9]], save=False)  # still usable
        assert d.shape == (4, 1)


def test_1024_cubed_grid_addressing_by_translation(gpu):
    """The largest grid the tests run: 1024^3 (4 GiB per field, element indices past 2^30, byte offsets past 2^32, a
    forward-term store of 26 GB whose slot offsets pass 2^32 elements).  No oracle finishes that in seconds, so the
    check is the translation property: a shot in the far corner of the big grid, inside a block of heterogeneous
    model, must give the seismograms, adjoint traces and gradient of the same shot in a 96^3 grid that holds just that
    block -- the wave cannot reach the small grid's near faces within ``nt`` steps (4 cells per step), and the far
    faces (with their absorbing border) coincide."""
    N, n, nt, npml = 1024, 96, 6, 8
    o = N - n
    rng = np.random.default_rng(7)
    blk = (2000.0 + 600.0 * rng.random((n, n, n))).astype(np.float32)
    src = np.array([[50, 48, 52], [70, 60, 44]], np.int32)
    rec = np.array([[46, 52, 49], [60, 55, 58], [80, 70, 66], [90, 88, 91]], np.int32)
    wav = rng.standard_normal((nt, 2)).astype(np.float32)
    res = rng.standard_normal((nt, 4)).astype(np.float32)
    with Engine((n, n, n), 10.0, 1e-3, nt, npml=npml, sigma_max=300.0) as e:
        d_s = e.forward(blk, (src, wav), rec, save=True)
        a_s = e.adjoint(res)
        g_s = e.gradient("slowness2")
    big = np.full((N, N, N), 2000.0, np.float32)
    big[o:, o:, o:] = blk
    with Engine((N, N, N), 10.0, 1e-3, nt, npml=npml, sigma_max=300.0) as e:
        d_b = e.forward(big, (src + o, wav), rec + o, save=True)
        a_b = e.adjoint(res)
        g_b = e.gradient("slowness2")
    del big
    assert np.all(np.isfinite(d_b)) and np.linalg.norm(d_s) > 0 and np.linalg.norm(g_s) > 0
    assert rel(d_b, d_s) < 1e-6 and rel(a_b, a_s) < 1e-6
    assert rel(g_b[o:, o:, o:], g_s) < 1e-6
    outside = g_b.copy()
    outside[o:, o:, o:] = 0.0
    assert not outside.any()  # nothing was written anywhere else in the 4 GiB image


@pytest.mark.parametrize("nt", [8, 10])
def test_16384_squared_grid_addressing_by_translation(gpu, nt):
    """The same property in 2-D at 16384^2 (1 GiB per field, ~80 k tiles of the fused kernel, the XCD-contiguous tile
    walk at that count); ``nt`` = two fused launches, and two plus a two-step tail."""
    N, n, npml = 16384, 160, 8
    o = N - n
    rng = np.random.default_rng(11)
    blk = (2000.0 + 600.0 * rng.random((n, n))).astype(np.float32)
    src = np.array([[80, 84], [100, 70]], np.int32)
    rec = np.array([[78, 88], [90, 95], [120, 110], [150, 152]], np.int32)
    wav = rng.standard_normal((nt, 2)).astype(np.float32)
    res = rng.standard_normal((nt, 4)).astype(np.float32)
    out = []
    for shape, model, off in (((n, n), blk, 0), ((N, N), None, o)):
        if model is None:
            model = np.full((N, N), 2000.0, np.float32)
            model[o:, o:] = blk
        with Engine(shape, 10.0, 1e-3, nt, npml=npml, sigma_max=300.0) as e:
            d = e.forward(model, (src + off, wav), rec + off, save=True)
            a = e.adjoint(res)
            g = e.gradient("slowness2")
            out.append((d, a, g, e.kernel_name))
    (d_s, a_s, g_s, _), (d_b, a_b, g_b, kern) = out
    assert kern == "step2d_fused"
    assert np.linalg.norm(d_s) > 0 and np.linalg.norm(g_s) > 0
    assert rel(d_b, d_s) < 1e-6 and rel(a_b, a_s) < 1e-6 and rel(g_b[o:, o:], g_s) < 1e-6
    g_b[o:, o:] = 0.0
    assert not g_b.any()


@pytest.mark.parametrize("maker", ["cfg3", "cfg5"])
def test_engine_pool_on_gpu_matches_single_engine(gpu, maker):
    """Shots overlapped on one GPU (three contexts, three host threads) == one after the other; 2-D (where the pool is
    the default) and 3-D (a pool of one by default, but nothing may depend on that)."""
    from full_waveform_inversion_amd import shots as sh

    w = workloads.cfg3(0.125, nshots=6) if maker == "cfg3" else workloads.cfg5(0.1875, nshots=9)
    wav = w.wavelet()

    def mk():
        return Engine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml, sigma_max=900.0)

    a = [sh.Shot(w.src_idx[i:i + 1], wav, w.rec_idx) for i in range(6)]
    b = [sh.Shot(w.src_idx[i:i + 1], wav, w.rec_idx) for i in range(6)]
    c0 = (0.97 * w.c_init).astype(np.float32)
    with mk() as e:
        sh.model_data(e, w.c.astype(np.float32), a)
        J1, g1 = sh.misfit_and_gradient(e, c0, a)
    with sh.EnginePool(mk, 3) as pool:
        sh.model_data(pool, w.c.astype(np.float32), b)
        J3, g3 = sh.misfit_and_gradient(pool, c0, b)
        with pytest.raises(FwiError):
            pool.primary.gradient_add_from(pool.primary)
    assert all(np.array_equal(x.d_obs, y.d_obs) for x, y in zip(a, b))
    assert abs(J3 - J1) < 1e-6 * J1 and rel(g3, g1) < 1e-6


@pytest.mark.parametrize("shape", [(40, 36, 44), (60, 52)])
def test_gpu_fp64_adjoint_identity(gpu, shape):
    """<F s, r> = <s, F^T r> on the GPU itself in fp64 (1e-12): forward and adjoint kernels,
    injection, sampling and their scalings are exact transposes -- no oracle involved."""
    rng = np.random.default_rng(4)
    nd = len(shape)
    c = 1600.0 + 1400.0 * rng.random(shape)
    dt = 0.7 * fo.cfl_dt(c.max(), 9.0, nd, 8)
    nt = 80
    src = np.stack([rng.integers(0, s, 3) for s in shape], 1)
    rec = np.stack([rng.integers(0, s, 13) for s in shape], 1)
    w = rng.standard_normal((nt, 3))
    r = rng.standard_normal((nt, 13))
    with Engine(shape, 9.0, dt, nt, order=8, npml=6, sigma_max=600.0, dtype="float64") as e:
        d = e.forward(c, (src, w), rec, save=False)
        a = e.adjoint(r, image=False)
    lhs, rhs = float(np.sum(d * r)), float(np.sum(w * a))
    assert abs(lhs - rhs) <= 1e-12 * max(abs(lhs), abs(rhs))


def test_random_small_configurations_fuzz(gpu):
    """Seeded fuzz over small odd grids (narrower than a tile, thinner than the stencil, borders as
    wide as the grid allows), all orders, 2-D / 3-D, fp32: forward + adjoint + gradient vs the C oracle."""
    fuzz_small_configurations(2026, 40)


def fuzz_small_configurations(seed, ncases, maxdim=41):
    """(also driven with other seeds and larger grids by tools/fuzz_soak.py)"""
    rng = np.random.default_rng(seed)
    for case in range(ncases):
        nd = int(rng.integers(2, 4))
        order = int(rng.choice([2, 4, 8]))
        shape = tuple(int(rng.integers(3, maxdim)) for _ in range(nd))
        if rng.random() < 0.5:  # make the float4 kernels eligible
            shape = shape[:-1] + (4 * int(rng.integers(1, 12)),)
        npml = int(rng.integers(0, max(1, min(shape) // 2)))
        nt = int(rng.choice([7, 8, 12, 16, 21]))
        c = 1500.0 + 2000.0 * rng.random(shape)
        dt = 0.6 * fo.cfl_dt(c.max(), 6.0, nd, order)
        nsrc, nrec = int(rng.integers(1, 4)), int(rng.integers(1, 9))
        src = np.stack([rng.integers(0, s, nsrc) for s in shape], 1)
        rec = np.stack([rng.integers(0, s, nrec) for s in shape], 1)
        w = rng.standard_normal((nt, nsrc))
        r = rng.standard_normal((nt, nrec))
        p = CPropagator(c, 6.0, dt, order, npml)
        d = p.forward(src, w, rec)
        a = p.adjoint(r)
        g = p.gradient()
        ck = int(rng.choice([0, 0, 4, 5]))
        poison_device_memory(64)
        with Engine(shape, 6.0, dt, nt, order=order, npml=npml, sigma_max=p.sigma_max, ckpt_interval=ck) as e:
            dg = e.forward(c, (src, w), rec, save=True)
            ag = e.adjoint(r)
            gg = e.gradient()
            kern = e.kernel_name
        tag = (seed, case, shape, order, npml, nt, ck, kern)
        # (+ FLOOR: a receiver the wave has not reached in nt steps records ~1e-50 in the fp64 oracle and 0 in fp32)
        assert np.linalg.norm(dg - d) < 2e-5 * np.linalg.norm(d) + FLOOR, tag
        assert np.linalg.norm(ag - a) < 2e-5 * np.linalg.norm(a) + FLOOR, tag
        assert np.linalg.norm(gg - g) < 5e-5 * np.linalg.norm(g) + FLOOR, tag


def test_random_option_combinations_fuzz(gpu, monkeypatch):
    """Seeded fuzz over the round-2 options on small odd grids: absorbing border (sponge / CPML with and without
    alpha), update form (standard / increment), forward-term store (native / bf16 where it applies), checkpointing
    where it applies, the two-steps-per-pass 3-D kernel for the store-free forward sweep -- forward + adjoint +
    gradient against the C oracle (the NumPy oracle's restatement for the bf16 store)."""
    fuzz_option_combinations(4052, 48, monkeypatch.setenv)


def fuzz_option_combinations(seed, ncases, setenv, maxdim=37):
    """(also driven with other seeds and larger grids by tools/fuzz_soak.py)"""
    rng = np.random.default_rng(seed)
    for case in range(ncases):
        nd = int(rng.integers(2, 4))
        order = int(rng.choice([2, 4, 8]))
        shape = tuple(int(rng.integers(6, maxdim)) for _ in range(nd))
        if rng.random() < 0.5:
            shape = shape[:-1] + (4 * int(rng.integers(2, 12)),)
        npml = int(rng.integers(0, max(1, min(shape) // 2)))
        nt = int(rng.choice([7, 8, 13, 16, 22]))
        c = 1500.0 + 2000.0 * rng.random(shape)
        h = 6.0
        dt = 0.6 * fo.cfl_dt(c.max(), h, nd, order)
        nsrc, nrec = int(rng.integers(1, 4)), int(rng.integers(1, 9))
        src = np.stack([rng.integers(0, s, nsrc) for s in shape], 1)
        rec = np.stack([rng.integers(0, s, nrec) for s in shape], 1)
        w = rng.standard_normal((nt, nsrc))
        r = rng.standard_normal((nt, nrec))
        abc = str(rng.choice(["sponge", "cpml"]))
        alpha = float(rng.choice([0.0, 25.0])) if abc == "cpml" else 0.0
        form = str(rng.choice(["standard", "increment"]))
        bf16 = nd == 3 and order == 8 and form == "standard" and (abc == "sponge" or npml == 0) and rng.random() < 0.4
        ck = int(rng.choice([0, 0, 5])) if not bf16 else 0
        pair = nd == 3 and rng.random() < 0.5
        stride = int(rng.choice([1, 1, 3])) if ck == 0 else 1
        setenv("FWI_STREAM_PAIR", "1" if pair else "0")
        # round 4: the tile shape of the 3-D stream kernel (8 rows = what grids past the Infinity Cache run) and the way
        # the time loop is submitted (one hipGraph per sweep) are part of the option space as well
        ty8 = bool(rng.random() < 0.5)
        setenv("FWI_STREAM_TY", "8" if ty8 else "4")
        launch = str(rng.choice(["stream", "graph"]))
        kw = dict(abc=abc, pml_alpha_max=alpha, image_stride=stride)
        if bf16:
            p = fo.Propagator(c, h, dt, order, npml, store_dtype="bf16", **kw)
        else:
            p = CPropagator(c, h, dt, order, npml, **kw)
        d = p.forward(src, w, rec)
        a = p.adjoint(r)
        g = p.gradient()
        poison_device_memory(64)
        with Engine(shape, h, dt, nt, order=order, npml=npml, sigma_max=p.sigma_max, ckpt_interval=ck,
                    update_form=form, store_dtype="bf16" if bf16 else "native", launch_mode=launch, **kw) as e:
            d0 = e.forward(c, (src, w), rec, save=False)   # (3-D, no border: the two-step kernel when enabled)
            dg = e.forward(None, (src, w), rec, save=True)
            ag = e.adjoint(r)
            gg = e.gradient()
            kern = e.kernel_name
        tag = (seed, case, shape, order, npml, nt, abc, alpha, form, bf16, ck, pair, stride, kern, ty8, launch)
        gtol = 5e-5
        if bf16:
            # fp32 and fp64 values of C L u that straddle a bf16 rounding boundary round apart by a whole bf16 ulp (the
            # Laplacian is a cancelling sum: its fp32 error is far above 1e-7 in smooth parts of the field), so the
            # engine's store and the oracle's restatement differ by a share of the quantisation effect itself
            pn = CPropagator(c, h, dt, order, npml, **kw)
            pn.forward(src, w, rec)
            pn.adjoint(r)
            gn = pn.gradient()
            quant = np.linalg.norm(g - gn) / max(np.linalg.norm(gn), 1e-300)
            gtol = 2e-4 + 0.5 * quant
            # ... but it must be the bf16 store that it reproduces: where the quantisation effect stands clear of the
            # fp32 error level, the engine's gradient lies nearer the bf16 oracle than the native one
            if quant > 2e-3:
                assert np.linalg.norm(gg - g) < np.linalg.norm(gg - gn), tag
        for x, ref, tol in ((d0, d, 2e-5), (dg, d, 2e-5), (ag, a, 2e-5), (gg, g, gtol)):
            assert np.linalg.norm(x - ref) < tol * np.linalg.norm(ref) + FLOOR, tag
