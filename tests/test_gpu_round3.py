"""Round-3 GPU tests: call-order state, NaN-preserving bf16 store."""
import numpy as np
import pytest

from full_waveform_inversion_amd import Engine, FwiError
from oracle import fwi_oracle as fo

pytestmark = pytest.mark.gpu


def _small_shot(shape, nt=40, order=8):
    rng = np.random.default_rng(3)
    c = 2000.0 + 500.0 * rng.random(shape)
    h = 10.0
    dt = 0.7 * fo.cfl_dt(c.max(), h, len(shape), order)
    src = np.array([[s // 2 for s in shape]])
    rec = np.array([[s // 3 for s in shape], [2 * s // 3 for s in shape]])
    wav = fo.ricker(nt, dt, 0.1 / dt / 8).astype(np.float32)
    return c, h, dt, src, rec, wav


@pytest.mark.parametrize("shape", [(40, 36, 44), (70, 90)])
def test_misfit_l2_after_an_adjoint_is_a_state_error(gpu, shape):
    """fwi_adjoint records its source-side series into the buffer the synthetics lived in: a misfit asked for after
    it must fail (FWI_ESTATE), not return a J built from adjoint data; a new forward makes it valid again."""
    c, h, dt, src, rec, wav = _small_shot(shape)
    with Engine(shape, h, dt, len(wav)) as e:
        d = e.forward(c, (src, wav), rec, save=False)
        J = e.misfit_l2(np.zeros_like(d))
        assert abs(J - 0.5 * float(np.sum(d.astype(np.float64) ** 2))) <= 1e-5 * J
        e.adjoint(d, image=False)
        with pytest.raises(FwiError) as ei:
            e.misfit_l2(np.zeros_like(d))
        assert ei.value.code == 3
        e.forward(None, (src, wav), rec, save=False)
        assert abs(e.misfit_l2(np.zeros_like(d)) - J) <= 1e-6 * J


def test_bf16_store_keeps_a_nan_a_nan(gpu):
    """A forward run that blows up (dt twice the CFL limit) must not come out of the bf16 forward-term store as a
    finite-looking gradient: the conversion is the plain cast (v_cvt_pk_bf16_f32), which keeps every NaN."""
    shape = (40, 36, 64)
    c, h, dt, src, rec, wav = _small_shot(shape, nt=400)
    dt_bad = 2.0 * fo.cfl_dt(c.max(), h, 3, 8)
    with Engine(shape, h, dt_bad, len(wav), store_dtype="bf16") as e:
        d = e.forward(c, (src, wav), rec, save=True)
        assert not np.isfinite(d).all()          # the run did blow up
        e.adjoint(np.ones_like(d) * 1e-3)
        g = e.gradient()
    assert not np.isfinite(g).all()
    # and a healthy run through the same store stays finite and close to the native store's gradient
    with Engine(shape, h, dt, 60, store_dtype="bf16") as e, Engine(shape, h, dt, 60) as e2:
        out = []
        for eng in (e, e2):
            dd = eng.forward(c, (src, wav[:60]), rec, save=True)
            eng.adjoint(dd)
            out.append(eng.gradient())
    assert np.isfinite(out[0]).all()
    assert np.linalg.norm(out[0] - out[1]) < 2e-2 * np.linalg.norm(out[1])


def test_device_lbfgs_resumes_from_its_state_file(gpu, tmp_path):
    """The device-resident optimiser writes its state after every iteration (model, gradient and curvature pairs
    downloaded once); a run resumed from the file of iteration 2 ends where the uninterrupted run ended.  (fp32
    gradients carry float atomics in the injection, so "equal" is to round-off here; the host optimiser's resume
    is bit-exact, tests/test_host_logic.py.)"""
    import shutil
    from full_waveform_inversion_amd import shots as sh, workloads
    from full_waveform_inversion_amd.lbfgs import lbfgs, lbfgs_device, load_state
    w = workloads.cfg5(0.1875, nshots=3)  # 48^3 (two of its shots)
    wav = w.wavelet()
    shots = [sh.Shot(w.src_idx[i:i + 1], wav, w.rec_idx) for i in range(2)]
    x0 = w.c_init.astype(np.float32)
    ck, ck2 = str(tmp_path / "s.npz"), str(tmp_path / "s2.npz")
    kw = dict(maxiter=4, history=3, first_step=40.0, bounds=(1000.0, 5000.0))
    with Engine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml) as e:
        sh.model_data(e, w.c.astype(np.float32), shots)
        fg = lambda xs, gs: sh.misfit_and_gradient_device(e, xs, gs, shots)  # noqa: E731
        x_ref, f_ref, log_ref = lbfgs_device(e, fg, x0, checkpoint=ck,
                                             callback=lambda it, *_: it == 2 and shutil.copy(ck, ck2), **kw)
        assert load_state(ck2)["it"] == 2 and load_state(ck)["it"] == 4
        x2, f2, log2 = lbfgs_device(e, fg, None, resume=ck2, **kw)
        # and the host optimiser can pick the same file up
        x3, f3, log3 = lbfgs(lambda m: sh.misfit_and_gradient(e, m, shots), None, resume=ck2, dot=e.dot, **kw)
    assert [r["evals"] for r in log2] == [r["evals"] for r in log_ref] == [r["evals"] for r in log3]
    assert abs(f2 - f_ref) <= 1e-5 * f_ref and abs(f3 - f_ref) <= 1e-4 * f_ref
    assert np.linalg.norm(x2 - x_ref) <= 1e-6 * np.linalg.norm(x_ref)


@pytest.mark.parametrize("shape", [(133, 260), (70, 52)])
def test_small_tile_instantiations_of_the_fused_2d_kernel_are_bit_identical(gpu, monkeypatch, shape):
    """grids that make fewer than 256 tiles of 64^2 run step2d_fused on 32- or 16-point tiles.
    Every tile size gives the BITS of step2d_tile -- seismograms, F^T r and gradient, duplicate source nodes, points on
    tile seams -- and the default picks a small tile for these grids."""
    rng = np.random.default_rng(21)
    c = (1500.0 + 1500.0 * rng.random(shape)).astype(np.float32)
    dt = 0.7 * fo.cfl_dt(float(c.max()), 5.0, 2, 8)
    nt = 64
    src = np.array([[shape[0] // 2, shape[1] // 2], [3, 3], [31, 32], [31, 32], [16, 15]])
    rec = np.stack([rng.integers(0, s, 40) for s in shape], 1)
    rec[:4] = [[15, 16], [32, 31], [shape[0] - 1, shape[1] - 1], [0, 0]]
    wav = rng.standard_normal((nt, len(src))).astype(np.float32)
    res = rng.standard_normal((nt, len(rec))).astype(np.float32)
    out = {}
    for name in ("tile", "64", "32", "16", "auto"):
        monkeypatch.delenv("FWI_NO_FUSED2D", raising=False)
        monkeypatch.delenv("FWI_FUSED2D_TILE", raising=False)
        if name == "tile":
            monkeypatch.setenv("FWI_NO_FUSED2D", "1")
        elif name != "auto":
            monkeypatch.setenv("FWI_FUSED2D_TILE", name)
        with Engine(shape, 5.0, dt, nt, order=8, npml=10, sigma_max=900.0, kernel="stream") as e:
            d = e.forward(c, (src, wav), rec, save=True)
            a = e.adjoint(res)
            out[name] = (e.kernel_name, d, a, e.gradient())
    assert out["tile"][0] == "step2d_tile" and all(out[k][0] == "step2d_fused" for k in ("64", "32", "16", "auto"))
    for k in ("64", "32", "16", "auto"):
        assert np.array_equal(out[k][1], out["tile"][1]), k          # seismograms: the same bits
        assert np.array_equal(out[k][2], out["tile"][2]), k          # F^T r
        assert rel(out[k][3], out["tile"][3]) < 2e-6, k              # gradient: other pairing order of the imaging sum


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / np.linalg.norm(b))
