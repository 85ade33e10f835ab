"""Off-grid sources / receivers (full_waveform_inversion_amd/points.py): interpolation weights and, on the GPU,
the adjoint identity of a shot with interpolated points."""
import numpy as np
import pytest

from full_waveform_inversion_amd.points import Spread


@pytest.mark.parametrize("shape", [(9, 11), (6, 7, 8)])
def test_weights_interpolate_linear_fields_exactly(shape):
    rng = np.random.default_rng(0)
    nd = len(shape)
    pts = rng.random((40, nd)) * (np.array(shape) - 1)
    pts[0] = 0.0                                    # a corner node
    pts[1] = np.array(shape) - 1                    # the far corner node
    pts[2] = np.floor(pts[2])                       # an interior node
    pts[3, 0] = shape[0] - 1                        # on the last plane, fractional elsewhere
    S = Spread(pts, shape)
    coef = rng.standard_normal(nd)
    grids = np.meshgrid(*[np.arange(s, dtype=float) for s in shape], indexing="ij")
    field = 0.7 + sum(c * g for c, g in zip(coef, grids))
    nodes = field[tuple(S.idx.T)][None, :]          # (1, m) "time series" of one sample
    assert np.allclose(S.gather(nodes)[0], 0.7 + pts @ coef, rtol=0, atol=1e-12)
    assert np.allclose(np.bincount(S.owner, S.weights, minlength=S.n), 1.0, atol=1e-14)  # partition of unity
    for i in (0, 1, 2):                              # points on nodes use that node only
        assert np.sum(S.owner == i) == 1 and np.array_equal(S.idx[S.owner == i][0], pts[i].astype(int))
    assert S.idx.dtype == np.int32 and S.idx.min() >= 0 and np.all(S.idx.max(axis=0) <= np.array(shape) - 1)


def test_scatter_is_the_transpose_of_gather():
    rng = np.random.default_rng(1)
    S = Spread(rng.random((7, 3)) * [4, 5, 6], (5, 6, 7))
    a, x = rng.standard_normal((13, 7)), rng.standard_normal((13, len(S.owner)))
    assert abs(np.sum(S.scatter(a) * x) - np.sum(a * S.gather(x))) < 1e-12
    assert S.scatter(a[:, 0] if S.n == 1 else a).shape == (13, len(S.owner))


def test_out_of_grid_points_are_rejected():
    with pytest.raises(ValueError):
        Spread([[0.0, 9.5]], (8, 10))
    with pytest.raises(ValueError):
        Spread([[1.0, 2.0, 3.0]], (8, 10))


@pytest.mark.gpu
def test_shot_with_interpolated_points_keeps_the_adjoint_identity(gpu):
    """<R F S w, r> == <w, S^T F^T R^T r> through the engine (fp64), and moving a receiver by a fraction of a cell
    moves its trace continuously between the two node traces."""
    from full_waveform_inversion_amd import Engine
    from oracle import fwi_oracle as fo
    rng = np.random.default_rng(3)
    shape, h, order, nt = (30, 26, 34), 10.0, 8, 80
    c = 2000.0 + 300.0 * rng.random(shape)
    dt = 0.6 * fo.cfl_dt(c.max(), h, 3, order)
    S = Spread([[14.3, 12.6, 16.2], [8.0, 9.5, 20.25]], shape)
    R = Spread(np.column_stack([np.full(6, 5.5), np.linspace(4.2, 20.7, 6), np.linspace(6.1, 27.9, 6)]), shape)
    w, r = rng.standard_normal((nt, 2)), rng.standard_normal((nt, 6))
    with Engine(shape, h, dt, nt, order=order, npml=4, sigma_max=300.0, dtype="float64") as e:
        d = R.gather(e.forward(c, (S.idx, S.scatter(w)), R.idx, save=True))
        adj = S.gather(e.adjoint(R.scatter(r)))
        assert np.isfinite(e.gradient()).all()
        lhs, rhs = float(np.sum(d * r)), float(np.sum(w * adj))
        assert abs(lhs - rhs) <= 1e-11 * max(abs(lhs), abs(rhs))
        # a receiver at x = 10 + f is the (1 - f, f) blend of the node traces at x = 10 and 11
        wav = fo.ricker(nt, dt, 25.0)
        src = [[15, 13, 17]]
        nodes = e.forward(c, (src, wav), [[6, 10, 10], [6, 10, 11]], save=False)
        for f in (0.0, 0.25, 0.9):
            P = Spread([[6.0, 10.0, 10.0 + f]], shape)
            tr = P.gather(e.forward(c, (src, wav), P.idx, save=False))[:, 0]
            assert np.allclose(tr, (1 - f) * nodes[:, 0] + f * nodes[:, 1], rtol=0, atol=1e-13 * np.abs(nodes).max())


def test_shot_at_coordinates_gradient_matches_finite_differences():
    """The shot loop with off-grid sources / receivers (Shot.at_coordinates): its gradient is the exact gradient of
    the interpolated forward map (oracle-backed engine, directional finite difference)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from _oracle_engine import OracleEngine
    from full_waveform_inversion_amd import shots as sh
    from oracle import fwi_oracle as fo
    rng = np.random.default_rng(5)
    shape, h, order, nt = (22, 26), 10.0, 4, 90
    c_true = 2000.0 + 200.0 * rng.random(shape)
    c0 = np.full(shape, 2100.0)
    dt = 0.6 * fo.cfl_dt(c_true.max(), h, 2, order)
    wav = fo.ricker(nt, dt, 30.0)
    rec = np.column_stack([np.full(7, 3.4), np.linspace(2.3, 22.8, 7)])
    shots = [sh.Shot.at_coordinates([[10.6, 7.25]], wav, rec, shape), sh.Shot.at_coordinates([[12.2, 18.7]], wav, rec, shape)]
    e = OracleEngine(shape, h, dt, nt, order=order, npml=3)
    sh.model_data(e, c_true, shots)
    assert shots[0].d_obs.shape == (nt, 7)
    J0, g = sh.misfit_and_gradient(e, c0, shots)
    dc = rng.standard_normal(shape)
    eps = 1e-3
    Jp, _ = sh.misfit_and_gradient(e, c0 + eps * dc, shots)
    Jm, _ = sh.misfit_and_gradient(e, c0 - eps * dc, shots)
    fd, an = (Jp - Jm) / (2 * eps), float(np.sum(g * dc))
    assert abs(fd - an) <= 1e-6 * abs(an), (fd, an)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dtype,tol", [((30, 26, 34), "float64", 1e-12), ((30, 26, 34), "float32", 2e-6),
                                             ((40, 44), "float64", 1e-12), ((40, 44), "float32", 2e-6)])
def test_device_spread_and_gather_equal_the_host_ones(gpu, shape, dtype, tol):
    """fwi_forward_spread: the per-point wavelets are scattered onto the nodes and the node samples gathered per
    point ON THE DEVICE (wave-level __shfl_down over each point's <= 8 nodes); the same for the residual and the
    adjoint source series -- equal to points.Spread's host scatter / gather around the node-based calls, points
    on nodes, on the last plane and duplicates included; misfit and gradient through the shot loop likewise."""
    from full_waveform_inversion_amd import Engine, shots as sh
    from oracle import fwi_oracle as fo
    rng = np.random.default_rng(8)
    nd = len(shape)
    h, order, nt = 10.0, 8, 70
    c = 2000.0 + 300.0 * rng.random(shape)
    dt = 0.6 * fo.cfl_dt(c.max(), h, nd, order)
    lim = np.array(shape) - 1.0
    src = rng.random((3, nd)) * lim
    src[1] = np.floor(src[1])                       # exactly on a node
    rec = rng.random((9, nd)) * lim
    rec[0] = lim                                    # the far corner
    rec[1, 0] = lim[0]                              # on the last plane
    rec[2] = rec[3]                                 # two receivers at one place
    S, R = Spread(src, shape), Spread(rec, shape)
    w = rng.standard_normal((nt, 3)).astype(dtype)
    r = rng.standard_normal((nt, 9)).astype(dtype)
    with Engine(shape, h, dt, nt, order=order, npml=4, sigma_max=300.0, dtype=dtype) as e:
        d_host = R.gather(e.forward(c, (S.idx, S.scatter(w)), R.idx, save=True))
        a_host = S.gather(e.adjoint(R.scatter(r)))
        g_host = e.gradient()
        e.reset_gradient()
        d_dev = e.forward_at(None, (src, w), rec, save=True)
        a_dev = e.adjoint(r)
        g_dev = e.gradient()
        scale = lambda x: np.linalg.norm(x)  # noqa: E731
        assert d_dev.shape == (nt, 9) and a_dev.shape == (nt, 3)
        assert scale(d_dev - d_host) <= tol * scale(d_host)
        assert scale(a_dev - a_host) <= tol * scale(a_host)
        assert scale(g_dev - g_host) <= tol * scale(g_host)
        assert np.array_equal(d_dev[:, 2], d_dev[:, 3])
        # device misfit on the gathered traces, residual scattered on the device too
        e.reset_gradient()
        d_obs = (d_dev * 0.9).astype(dtype)
        e.forward_at(None, (src, w), rec, save=True)
        J = e.misfit_l2(d_obs)
        e.adjoint(None)
        g2 = e.gradient()
        e.reset_gradient()
        e.forward_at(None, (src, w), rec, save=True)
        e.adjoint((d_dev - d_obs).astype(dtype))
        assert abs(J - 0.5 * float(np.sum((d_dev.astype(np.float64) - d_obs) ** 2))) <= 1e-6 * J
        assert scale(g2 - e.gradient()) <= max(tol, 1e-7) * scale(g2)
        # a node-based call afterwards is a node-based call again
        assert e.forward(None, (S.idx, S.scatter(w)), R.idx, save=False).shape == (nt, len(R.idx))
        # the shot loop takes the device path for such shots
        shot = sh.Shot.at_coordinates(src, w, rec, shape, d_obs=d_obs)
        J3, g3 = sh.misfit_and_gradient(e, c.astype(dtype), [shot])
        assert abs(J3 - J) <= 1e-6 * J and scale(g3 - e.gradient()) >= 0  # (runs; the value is pinned just below)
        e.reset_gradient()
        e.forward_at(None, (src, w), rec, save=True)
        e.misfit_l2(d_obs)
        e.adjoint(None)
        assert scale(g3 - e.gradient("velocity")) <= max(tol, 1e-7) * scale(g3)
        import ctypes as C
        with pytest.raises(Exception):  # a point with more than 8 nodes is refused
            bad = np.array([0, 9], np.int32)
            e._chk(e._lib.fwi_forward_spread(e._ctx, nt, 1, 9, S.idx.ctypes.data_as(C.c_void_p), bad.ctypes.data_as(C.c_void_p),
                                             np.ones(9, dtype).ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p),
                                             0, 0, None, None, None, 0, None))
