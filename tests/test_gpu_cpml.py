"""The convolutional PML (fwi_config.abc = FWI_ABC_CPML) on the GPU against the CPU oracle: seismograms, F^T r and
gradient; the exact adjoint identity in fp64; absorption against the sponge at equal border width."""
import numpy as np
import pytest

from full_waveform_inversion_amd import Engine, FwiError, workloads
from oracle import fwi_oracle as fo
from oracle.c_oracle import CPropagator

pytestmark = pytest.mark.gpu
TOL32, TOL64 = 1e-5, 1e-10


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / np.linalg.norm(b))


CASES = [  # shape, order, npml, alpha, dtype, tol, kernel, engine options
    ((40, 36, 44), 8, 6, 30.0, "float32", TOL32, "step3d_stream", {}),
    ((33, 29, 50), 4, 5, 0.0, "float32", TOL32, "step3d_stream", {}),
    ((36, 30, 48), 8, 8, 30.0, "float32", TOL32, "step3d_stream", {}),    # npml, nx multiples of 4: 16-byte lanes on x too
    ((30, 26, 24), 8, 8, 0.0, "float32", TOL32, "step3d_stream", {}),     # ... with the two x borders back to back
    ((72, 96), 8, 12, 40.0, "float32", TOL32, "step2d_tile", {}),
    ((40, 36, 44), 8, 6, 30.0, "float64", TOL64, "step3d_stream", {}),
    ((22, 9, 30), 2, 5, 20.0, "float64", TOL64, "step3d_stream", {}),     # ny < 2 npml: the y borders overlap
    ((40, 36, 44), 8, 6, 30.0, "float32", TOL32, "step3d_stream", {"update_form": "increment"}),
    ((40, 36, 44), 8, 6, 30.0, "float32", TOL32, "step_point", {"kernel": "point"}),
    ((40, 36, 44), 8, 6, 30.0, "float32", TOL32, "step3d_stream", {"ckpt_interval": 16}),  # snapshots hold psi, zeta
    ((40, 36, 44), 8, 6, 30.0, "float64", TOL64, "step3d_stream", {"ckpt_interval": 7, "update_form": "standard"}),
    ((70, 90), 8, 8, 40.0, "float32", TOL32, "step2d_tile", {"ckpt_interval": 10}),
    ((70, 90), 8, 8, 40.0, "float32", TOL32, "step2d_tile", {}),          # 2-D: one step per launch with the CPML
    ((70, 91), 2, 7, 0.0, "float32", TOL32, "step2d_tile", {}),
    ((70, 90), 8, 8, 40.0, "float64", TOL64, "step_point", {}),
    # 3-D grids whose x border runs inside step3d_stream's lanes (stream_xpml_supported: npml, nx multiples of 4)
    ((40, 36, 64), 8, 8, 30.0, "float32", TOL32, "step3d_stream", {"ckpt_interval": 16}),
    ((20, 18, 300), 8, 8, 30.0, "float32", TOL32, "step3d_stream", {}),           # two x tiles of 152 / 148 columns
    ((24, 16, 256), 8, 16, 20.0, "float32", TOL32, "step3d_stream", {}),          # full 256-column tiles (FULL path)
    ((30, 24, 40), 8, 12, 0.0, "float32", TOL32, "step3d_stream", {"zchunk": 7}), # several z chunks per tile
    ((40, 36, 64), 8, 8, 30.0, "float32", TOL32, "step3d_stream", {"update_form": "increment"}),  # ... in increment form
    # ... border widths that are even but not multiples of 4: one lane per side astride the border's inner edge (masked stores)
    ((40, 36, 64), 8, 10, 30.0, "float32", TOL32, "step3d_stream", {}),
    ((36, 30, 300), 8, 14, 0.0, "float32", TOL32, "step3d_stream", {"update_form": "increment", "ckpt_interval": 16}),
    ((26, 22, 256), 8, 6, 20.0, "float32", TOL32, "step3d_stream", {}),           # (FULL path)
    # thin grids with several z chunks per tile: chunk seams between / inside the borders' shells
    ((64, 20, 48), 8, 8, 30.0, "float32", TOL32, "step3d_stream", {"zchunk": 32}),   # a seam between the two borders
    ((56, 18, 44), 8, 8, 0.0, "float32", TOL32, "step3d_stream", {"zchunk": 56}),    # one chunk holds both
    ((70, 17, 40), 8, 12, 25.0, "float32", TOL32, "step3d_stream", {"zchunk": 35, "ckpt_interval": 20}),
    # line launches (fwi_pml.hip, pml_line): long borders (several blocks per line), O(2) / O(4) windows, fp64 lanes,
    # borders that nearly meet (one merged segment per line: n < 2 npml + 3 r), the point kernel around them
    ((72, 40, 36), 8, 20, 25.0, "float32", TOL32, "step3d_stream", {}),
    ((31, 27, 40), 4, 9, 10.0, "float64", TOL64, "step3d_stream", {}),
    ((26, 23, 33), 2, 7, 0.0, "float32", TOL32, "step3d_stream", {}),
    ((34, 33, 32), 8, 11, 30.0, "float64", TOL64, "step_point", {"update_form": "increment"}),  # ny = 33 < 2 * 11 + 12
    ((34, 35, 28), 8, 11, 30.0, "float32", TOL32, "step_point", {"kernel": "point", "ckpt_interval": 9}),
    # grids the fused 2-D kernel takes WITH the border recursion inside the launch (fused2d_cpml_supported): 70 steps
    # = 68 in 4-step launches + 2 through the slab path, on the same memory variables
    ((192, 256), 8, 40, 40.0, "float32", TOL32, "step2d_fused", {}),            # cfg2's border width, 48-cell images
    ((176, 240), 8, 16, 30.0, "float32", TOL32, "step2d_fused", {}),            # last tiles 48 wide
    ((150, 216), 8, 6, 0.0, "float32", TOL32, "step2d_fused", {}),              # high borders off the 16-byte groups,
                                                                                # last tiles exactly npml + 16 wide
    ((192, 256), 8, 24, 40.0, "float32", TOL32, "step2d_fused", {"ckpt_interval": 16}),
    ((1130, 1070), 8, 20, 30.0, "float32", TOL32, "step2d_fused", {}),          # 306 tiles: more than one round of
                                                                                # workgroups, border tiles first
    # whole tiles with an overlap seam in the middle of each axis (fused2d_origin): a last tile narrower than npml + 16
    # no longer sends the grid to the slab path
    ((149, 216), 8, 6, 0.0, "float32", TOL32, "step2d_fused", {}),              # seams of 43 and 40 cells
    ((200, 300), 8, 40, 35.0, "float32", TOL32, "step2d_fused", {}),            # ... with cfg2's border width
    ((333, 131), 8, 24, 20.0, "float32", TOL32, "step2d_fused", {"ckpt_interval": 12}),  # odd sizes: origins rounded to 4
    ((130, 250), 8, 16, 30.0, "float32", TOL32, "step2d_fused", {}),            # two tiles along z: both hold a border
    ((60, 216), 8, 6, 0.0, "float32", TOL32, "step2d_tile", {}),                # one tile along z: slab path
]


# 3-D stream-kernel cases again with 8-row tiles (FWI_STREAM_TY=8): the shape stream_default_tuning picks for every
# grid past the Infinity Cache (>= 272^3) -- none of which an oracle finishes in seconds, so the hook brings the tile
# shape to the oracle-sized grids instead (missing #1).  Same bar.
CASES_TY8 = [c for c in CASES if c[6] == "step3d_stream" and len(c[0]) == 3]


@pytest.mark.parametrize("shape,order,npml,alpha,dtype,tol,kern,kw", CASES_TY8)
def test_cpml_vs_oracle_with_8_row_tiles(gpu, monkeypatch, shape, order, npml, alpha, dtype, tol, kern, kw):
    monkeypatch.setenv("FWI_STREAM_TY", "8")
    test_cpml_vs_oracle(gpu, shape, order, npml, alpha, dtype, tol, kern, kw)


@pytest.mark.parametrize("shape,order,npml,alpha,dtype,tol,kern,kw", CASES)
def test_cpml_vs_oracle(gpu, shape, order, npml, alpha, dtype, tol, kern, kw):
    rng = np.random.default_rng(11)
    c = 2000.0 + 600.0 * rng.random(shape)
    h = 10.0
    dt = 0.7 * fo.cfl_dt(c.max(), h, len(shape), order)
    nt = 70
    src = np.array([[s // 2 for s in shape], [2] + [s // 3 for s in shape[1:]]])  # one source inside the border
    rec = np.array([[1] + [s // 3 for s in shape[1:]], [s // 2 + 2 for s in shape], [s - 2 for s in shape]])
    wav = np.stack([fo.ricker(nt, dt, 0.12 / dt / 8), 0.5 * fo.ricker(nt, dt, 0.12 / dt / 6)], 1)
    p = fo.Propagator(c, h, dt, order, npml, abc="cpml", pml_alpha_max=alpha)
    d = p.forward(src, wav, rec)
    r = d * rng.uniform(0.5, 1.5, size=(1, len(rec))) + 0.1 * np.roll(d, 2, axis=0)
    a = p.adjoint(r)
    g = p.gradient()
    with Engine(shape, h, dt, nt, order=order, npml=npml, sigma_max=p.sigma_max, dtype=dtype, abc="cpml",
                pml_alpha_max=alpha, **kw) as e:
        dg = e.forward(c, (src, wav), rec, save=True)
        assert e.kernel_name == kern
        ag = e.adjoint(r)
        gg = e.gradient()
        dg2 = e.forward(None, (src, wav), rec, save=False)  # the memory variables restart from zero
    assert np.array_equal(dg, dg2) or rel(dg2, dg) < 1e-6
    # (flat tolerance on all three since round 4: profiles/r03_parity.json measures <= 6e-6 on every fp32 case)
    assert rel(dg, d) < tol and rel(ag, a) < tol and rel(gg, g) < tol, (rel(dg, d), rel(ag, a), rel(gg, g))


@pytest.mark.parametrize("stream_ty", [None, "8"])
@pytest.mark.parametrize("kw", [{}, {"update_form": "increment"}, {"dtype": "float64"}, {"kernel": "point"}])
def test_cpml_line_launches_equal_the_slab_path(gpu, monkeypatch, kw, stream_ty):
    """The same shot with the z / y borders' term handed over by the line launch (default; the x border then in the
    stream kernel's lanes) and with all three axes as slab phases around the step kernel (FWI_NO_PML_LINES=1):
    seismograms, F^T r and gradient agree to round-off.  4- and 8-row tiles; the point kernel takes the term too."""
    if stream_ty:
        monkeypatch.setenv("FWI_STREAM_TY", stream_ty)
    rng = np.random.default_rng(8)
    shape, npml, nt = (44, 40, 48), 8, 60
    c = 1900.0 + 700.0 * rng.random(shape)
    h, order = 10.0, 8
    dt = 0.7 * fo.cfl_dt(c.max(), h, 3, order)
    src = np.array([[2, 20, 24], [22, 3, 24]])
    rec = np.array([[1, 18, 20], [40, 36, 5], [22, 20, 24]])
    wav = np.stack([fo.ricker(nt, dt, 0.12 / dt / 8), 0.5 * fo.ricker(nt, dt, 0.12 / dt / 6)], 1)
    out = []
    for env in (None, "1"):
        if env:
            monkeypatch.setenv("FWI_NO_PML_LINES", env)
        with Engine(shape, h, dt, nt, order=order, npml=npml, sigma_max=800.0, abc="cpml", pml_alpha_max=25.0, **kw) as e:
            d = e.forward(c, (src, wav), rec, save=True)
            a = e.adjoint(0.7 * d + 0.1 * np.roll(d, 3, axis=0))
            out.append((d, a, e.gradient()))
    tol = 1e-12 if kw.get("dtype") == "float64" else 1e-5  # (fp32: the two forms sum the second difference differently)
    for x, y in zip(*out):
        assert rel(x, y) < tol


@pytest.mark.parametrize("shape", [(40, 36, 44), (60, 52)])
def test_cpml_adjoint_identity_fp64_on_gpu(gpu, shape):
    rng = np.random.default_rng(4)
    c = 1800.0 + 900.0 * rng.random(shape)
    h, order, npml, nt = 10.0, 8, 6, 90
    dt = 0.7 * fo.cfl_dt(c.max(), h, len(shape), order)
    src = np.stack([rng.integers(0, s, 3) for s in shape], 1)
    rec = np.stack([rng.integers(0, s, 5) for s in shape], 1)
    w, r = rng.standard_normal((nt, 3)), rng.standard_normal((nt, 5))
    with Engine(shape, h, dt, nt, order=order, npml=npml, sigma_max=700.0, dtype="float64", abc="cpml",
                pml_alpha_max=35.0) as e:
        d = e.forward(c, (src, w), rec, save=False)
        a = e.adjoint(r, image=False)
    lhs, rhs = float(np.sum(d * r)), float(np.sum(w * a))
    assert abs(lhs - rhs) <= 1e-11 * max(abs(lhs), abs(rhs))


def test_cpml_absorbs_far_better_than_the_sponge_on_gpu(gpu):
    """cfg2-like homogeneous 2-D run, 10-cell border, fp32: residual reflection against a large-grid reference."""
    h, order, c0, n, big, npml, f0 = 10.0, 8, 2000.0, 128, 420, 10, 12.0
    dt = 0.7 * fo.cfl_dt(c0, h, 2, order)
    nt = 600
    w = fo.ricker(nt, dt, f0).astype(np.float32)

    def run(nn, abc, width):
        off = (nn - n) // 2
        rec = np.array([[off + npml + 2, off + x] for x in range(npml + 2, n - npml - 2, 4)])
        with Engine((nn, nn), h, dt, nt, order=order, npml=width, abc=abc) as e:
            return e.forward(np.full((nn, nn), c0, np.float32), (np.array([[off + n // 2, off + n // 2]]), w), rec,
                             save=False)

    ref = run(big, "sponge", 0)
    e_sponge, e_cpml = rel(run(n, "sponge", npml), ref), rel(run(n, "cpml", npml), ref)
    assert e_sponge > 0.05 and e_cpml < 2e-3


def _cpml_vs_c_oracle(w, nt, alpha, kern, **kw):
    wav = w.wavelet(np.float64)[:nt]
    p = CPropagator(w.c, w.h, w.dt, w.order, w.npml, abc="cpml", pml_alpha_max=alpha)
    d = p.forward(w.src_idx, wav, w.rec_idx, save=True)
    r = 0.7 * d + 0.2 * np.roll(d, 3, axis=0)
    a = p.adjoint(r)
    g = p.gradient("velocity")
    p.q_store = None
    with Engine(w.shape, w.h, w.dt, nt, order=w.order, npml=w.npml, sigma_max=p.sigma_max, abc="cpml",
                pml_alpha_max=alpha, **kw) as e:
        dg = e.forward(w.c, (w.src_idx, wav), w.rec_idx, save=True)
        assert e.kernel_name == kern
        ag = e.adjoint(r)
        gg = e.gradient("velocity")
    assert rel(dg, d) < TOL32, rel(dg, d)
    assert rel(ag, a) < TOL32, rel(ag, a)
    assert rel(gg, g) < TOL32, rel(gg, g)


def test_cpml_configs1_at_full_size_vs_c_oracle(gpu):
    """configs[1] as BASELINE.json words it -- 2-D 1024^2 layered model, O(8) stencil + PML (here: the convolutional PML,
    npml 40, inside the fused 4-step launch) -- at full grid size, 1000 of its 2000 steps: seismograms, F^T r and the
    gradient of a shared residual against the C oracle."""
    _cpml_vs_c_oracle(workloads.cfg2(1.0), 1000, 3.14159 * 15.0, "step2d_fused")


@pytest.mark.parametrize("stream_ty,kw", [(None, {}), ("8", {}), ("8", {"update_form": "increment"}),
                                          ("place", {}), ("place", {"update_form": "increment"})])
def test_cpml_3d_lines_and_lanes_at_size_vs_c_oracle(gpu, monkeypatch, stream_ty, kw):
    """3-D 160^3 heterogeneous model, npml 16, 400 steps: the x border in step3d_stream's lanes, the z and y borders'
    term from the line launch (several workgroup tiles, all z chunks, two segments per line) against the C oracle --
    with the tuned tile shape and with the 8-row tiles of the HBM-regime grids."""
    if stream_ty == "place":  # the placement search (fwi_api.hip tune_placement) forced onto this grid: same numbers
        monkeypatch.setenv("FWI_PLACEMENT_TUNE", "force")
    elif stream_ty:
        monkeypatch.setenv("FWI_STREAM_TY", stream_ty)
    w = workloads.cfg5(0.625, nshots=1)
    w.npml = 16
    _cpml_vs_c_oracle(w, 400, 3.14159 * 10.0, "step3d_stream", **kw)


@pytest.mark.parametrize("shape,npml", [((130, 250), 16), ((1130, 1070), 40)])
def test_cpml_fused_launch_is_reproducible_run_to_run(gpu, shape, npml):
    """Tiles read the memory variables of border cells in their halo, which a neighbouring tile owns: the launch writes
    into a second set of arrays (swapped afterwards), and at the overlap seam only the owning tile stores -- a tile that
    stored the rows it merely holds made this shot differ from run to run (4e-3, found by the oracle test's repeat)."""
    rng = np.random.default_rng(3)
    c = (1900.0 + 800.0 * rng.random(shape)).astype(np.float32)
    h, order, nt = 10.0, 8, 72
    dt = 0.7 * fo.cfl_dt(float(c.max()), h, 2, order)
    src = np.array([[3, shape[1] // 2], [shape[0] // 2, 2]])
    rec = np.array([[1, 5], [shape[0] // 2, shape[1] // 2], [shape[0] - 2, shape[1] - 3], [66, 3], [3, shape[1] - 2]])
    wav = np.stack([fo.ricker(nt, dt, 0.12 / dt / 8)] * 2, 1).astype(np.float32)
    with Engine(shape, h, dt, nt, order=order, npml=npml, sigma_max=900.0, abc="cpml", pml_alpha_max=25.0) as e:
        runs = [e.forward(c if i == 0 else None, (src, wav), rec, save=(i % 2 == 0)) for i in range(5)]
        assert e.kernel_name == "step2d_fused"
    for r in runs[1:]:
        assert np.array_equal(r, runs[0])


@pytest.mark.parametrize("stream_ty", [None, "8"])
@pytest.mark.parametrize("shape,npml,kw", [((96, 72, 128), 16, {}), ((60, 52, 64), 8, {"update_form": "increment"}),
                                           ((41, 37, 50), 9, {"dtype": "float64"})])
def test_cpml_3d_is_reproducible_run_to_run(gpu, monkeypatch, shape, npml, kw, stream_ty):
    """Lanes + line launch (or slabs + lines): a shot repeated on the same context, with and without the forward-term
    store, returns the same bits -- the two segments of a line, the two axes and the step kernel never touch a cell at
    the same time."""
    if stream_ty:
        monkeypatch.setenv("FWI_STREAM_TY", stream_ty)
    rng = np.random.default_rng(6)
    c = 1900.0 + 800.0 * rng.random(shape)
    h, order, nt = 10.0, 8, 60
    dt = 0.7 * fo.cfl_dt(c.max(), h, 3, order)
    src = np.array([[2, shape[1] // 2, shape[2] // 2], [shape[0] // 2, 3, shape[2] - 3]])
    rec = np.array([[1, 5, 7], [shape[0] // 2, shape[1] // 2, shape[2] // 2], [shape[0] - 2, shape[1] - 3, 2]])
    wav = np.stack([fo.ricker(nt, dt, 0.12 / dt / 8)] * 2, 1)
    with Engine(shape, h, dt, nt, order=order, npml=npml, sigma_max=900.0, abc="cpml", pml_alpha_max=25.0, **kw) as e:
        runs = [e.forward(c if i == 0 else None, (src, wav), rec, save=(i % 2 == 1)) for i in range(4)]
        a1 = e.adjoint(runs[-1])
        g1 = e.gradient()
        e.forward(None, (src, wav), rec, save=True)
        e.reset_gradient()
        a2 = e.adjoint(runs[-1])
        g2 = e.gradient()
    for r in runs[1:]:
        assert np.array_equal(r, runs[0])
    assert np.array_equal(a1, a2) and np.array_equal(g1, g2)


@pytest.mark.parametrize("shape,npml,nt", [((192, 256), 40, 120), ((150, 216), 6, 101)])
def test_cpml_inside_the_fused_launch_equals_the_slab_path(gpu, monkeypatch, shape, npml, nt):
    """The same shot through step2d_fused with the border recursion inside the launch and through step2d_tile +
    slab kernels (FWI_NO_FUSED2D_CPML=1): seismograms, F^T r and gradient agree to fp32 round-off."""
    rng = np.random.default_rng(5)
    c = (1800.0 + 900.0 * rng.random(shape)).astype(np.float32)
    h, order = 10.0, 8
    dt = 0.7 * fo.cfl_dt(float(c.max()), h, 2, order)
    src = np.array([[3, shape[1] // 2], [shape[0] // 2, 2]])
    rec = np.array([[1, shape[1] // 2 - 3], [8, shape[1] // 2 + 6], [shape[0] // 2 + 5, 1], [shape[0] // 2 - 7, 9]])
    f0 = 1.0 / (24.0 * dt)  # 24 steps per period, onset well inside the run
    wav = np.stack([fo.ricker(nt, dt, f0, t0=1.2 / f0), fo.ricker(nt, dt, 0.7 * f0, t0=1.0 / f0)], 1).astype(np.float32)
    out = []
    for env in (None, "1"):
        if env:
            monkeypatch.setenv("FWI_NO_FUSED2D_CPML", env)
        with Engine(shape, h, dt, nt, order=order, npml=npml, sigma_max=900.0, abc="cpml", pml_alpha_max=30.0) as e:
            d = e.forward(c, (src, wav), rec, save=True)
            kern = e.kernel_name
            a = e.adjoint(d)
            out.append((kern, d, a, e.gradient()))
    assert out[0][0] == "step2d_fused" and out[1][0] == "step2d_tile"
    for i in (1, 2, 3):  # (two summation orders of the same recursion, 100+ steps deep in the border)
        assert rel(out[0][i], out[1][i]) < 1e-5, (i, rel(out[0][i], out[1][i]))


def test_cpml_configuration_errors(gpu):
    with pytest.raises(FwiError) as ei:
        Engine((32, 32, 32), 10.0, 1e-3, 8, npml=4, sigma_max=100.0, abc="cpml", store_dtype="bf16")
    assert ei.value.code == 1
    with pytest.raises(FwiError):
        Engine((32, 32), 10.0, 1e-3, 8, npml=4, sigma_max=100.0, abc="cpml", pml_alpha_max=-1.0)
    with Engine((32, 32), 10.0, 1e-3, 8, npml=0, abc="cpml") as e:  # no border: nothing to absorb with, plain run
        d = e.forward(np.full((32, 32), 2000.0, np.float32), ([[16, 16]], np.ones(8, np.float32)), [[8, 8]])
        assert d.shape == (8, 1) and e.kernel_name == "step2d_fused"
