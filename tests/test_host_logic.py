"""Host-side logic on CPU: workloads, shot sharding, the stdlib control plane and the exchange over it
(world_size 2 and 3, also under torch.distributed.run), L-BFGS."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from full_waveform_inversion_amd import shots as sh, workloads
from full_waveform_inversion_amd.lbfgs import lbfgs

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _oracle_engine import OracleEngine  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_covers_every_shot_once():
    for n in (0, 1, 5, 32, 64):
        for world in (1, 2, 3, 8):
            parts = [sh.partition_shots(n, r, world) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1
    with pytest.raises(ValueError):
        sh.partition_shots(4, 2, 2)


@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg3", "cfg4", "cfg5"])
def test_workloads_are_deterministic_and_stable(name):
    a, b = workloads.CONFIGS[name](0.125), workloads.CONFIGS[name](0.125)
    assert np.array_equal(a.c, b.c) and a.dt == b.dt and np.array_equal(a.rec_idx, b.rec_idx)
    from full_waveform_inversion_amd import cfl_dt
    assert a.dt < cfl_dt(a.c.max(), a.h, a.ndim, a.order)
    for idx in (a.src_idx, a.rec_idx):
        assert idx.min() >= 0 and all(idx[:, k].max() < a.shape[k] for k in range(a.ndim))


def test_full_size_configs_match_baseline_json():
    assert workloads.cfg1().shape == (256, 256) and workloads.cfg1().nt == 500 and workloads.cfg1().order == 2
    w2 = workloads.cfg2()
    assert w2.shape == (1024, 1024) and w2.nt == 2000 and w2.order == 8 and w2.npml > 0
    assert len(workloads.cfg3().src_idx) == 32
    w4 = workloads.cfg4()
    assert w4.shape == (256, 256, 256) and w4.nt == 1000 and w4.order == 8


def _serial_reference():
    w = workloads.cfg3(0.0625, nshots=5)
    wav = w.wavelet(np.float64)
    shots = [sh.Shot(w.src_idx[i:i + 1], wav, w.rec_idx) for i in range(len(w.src_idx))]
    e = OracleEngine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml)
    sh.model_data(e, w.c, shots)
    return sh.misfit_and_gradient(e, w.c_init, shots)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn_ranks(script, args, world, port=None, extra_env=None):
    port = port or _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2", **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", script)] + list(args),
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        o, _ = p.communicate(timeout=300)
        outs.append((p.returncode, o.decode()[-2000:]))
    return outs


@pytest.mark.parametrize("world", [2, 3])
def test_control_plane_collectives(tmp_path, world):
    """broadcast / barrier / allreduce (sum, max, min) / array sum / gather of the stdlib rendezvous."""
    out = str(tmp_path / "rdzv")
    for rc, o in _spawn_ranks("_rdzv_worker.py", [out], world):
        assert rc == 0, o
    assert all(os.path.exists("%s.rank%d" % (out, r)) for r in range(world))


def test_control_plane_under_torch_distributed_run(tmp_path):
    """The driver launches bench.py with `python -m torch.distributed.run`: its agent owns MASTER_PORT, the ranks
    must still find each other (ports MASTER_PORT + 1 ...), and no rank imports torch."""
    out = str(tmp_path / "rdzv")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_rdzv_worker.py"), out]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300,
                       env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert p.returncode == 0, p.stdout.decode()[-3000:]
    assert open(out + ".rank0").read() == "ok 0 2" and open(out + ".rank1").read() == "ok 1 2"


def test_control_plane_skips_a_foreign_listener_and_times_out_alone(monkeypatch):
    """A rank 0 of ANOTHER job (different MASTER_PORT token) on the same control port is not joined; a rank whose
    rank 0 never appears gives up with an error instead of hanging."""
    import threading
    from full_waveform_inversion_amd.rendezvous import Rendezvous, RendezvousError
    port = _free_port()
    monkeypatch.setenv("FWI_RDZV_PORT", str(_free_port()))  # both jobs are pointed at ONE explicit control port
    res = {}

    def other_job():  # listens on the control port (its own MASTER_PORT is port + 1), waits for a peer that never comes
        try:
            Rendezvous(0, 2, "127.0.0.1", port + 1, timeout=3.0)
        except RendezvousError as ex:
            res["other"] = str(ex)

    t = threading.Thread(target=other_job)
    t.start()
    with pytest.raises(RendezvousError, match="found no rank 0"):
        Rendezvous(1, 2, "127.0.0.1", port, timeout=1.5)  # meets the other job's listener, is refused
    t.join()
    assert "only 1 of 2 ranks joined" in res["other"]
    with pytest.raises(ValueError):
        Rendezvous(2, 2)


def test_control_ports_keep_away_from_master_ports_and_messages_are_capped():
    """jobs on one node get adjacent MASTER_PORTs -- their control windows must be disjoint and must
    not contain a launcher port; a peer that announces an absurd message length is an error, not an allocation."""
    import struct
    from full_waveform_inversion_amd import rendezvous as rz
    for mp in (29500, 29501, 29555, 12345, 65000):
        a, b = rz.control_port_base(mp), rz.control_port_base(mp + 1)
        assert 20000 <= a and a + 64 <= 29500 and abs(a - b) >= 64, (mp, a, b)
    s1, s2 = socket.socketpair()
    try:
        s1.sendall(struct.pack("!Q", rz.MAX_MESSAGE_BYTES + 1))
        with pytest.raises(rz.RendezvousError, match="limit"):
            rz._recv_msg(s2)
        rz._send_msg(s1, b"abc")
        assert rz._recv_msg(s2) == b"abc"
    finally:
        s1.close()
        s2.close()


def test_control_plane_bounded_wait_for_an_optional_phase():
    """`set_timeout`: after set-up a collective gives up on a silent peer within the new bound (bench.py's optional
    2-D scaling leg must never hold the ranks for the set-up timeout), and on a peer that has gone at once."""
    import socket
    import threading
    import time
    from full_waveform_inversion_amd.rendezvous import Rendezvous, RendezvousError
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    box = {}

    def peer():
        box["r1"] = Rendezvous(1, 2, "127.0.0.1", port, timeout=20.0)
        box["r1"].barrier()          # takes part in the first barrier, then stays silent

    t = threading.Thread(target=peer)
    t.start()
    r0 = Rendezvous(0, 2, "127.0.0.1", port, timeout=20.0)
    r0.barrier()
    t.join()
    r0.set_timeout(0.5)
    t0 = time.monotonic()
    with pytest.raises((RendezvousError, OSError)):
        r0.barrier()
    assert time.monotonic() - t0 < 5.0
    box["r1"].close()
    with pytest.raises((RendezvousError, OSError)):
        r0.allreduce([1.0])
    r0.close()


def test_no_torch_on_the_product_and_bench_path():
    """north_star: "host code stays in Python calling HIP through a thin ctypes C-ABI shim (no PyTorch)"."""
    import re
    files = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "tools", "run_config.py")]
    for dirpath, _, fs in os.walk(os.path.join(ROOT, "full_waveform_inversion_amd")):
        files += [os.path.join(dirpath, f) for f in fs if f.endswith(".py")]
    for f in files:
        assert not re.search(r"^\s*(import|from)\s+torch\b", open(f).read(), re.M), f


def test_the_package_never_reaches_the_oracle():
    """oracle/ is test infrastructure: nothing under the package may import it, in source or at run time (bench.py
    does, for its cpu_baseline leg only)."""
    import re
    import subprocess
    import sys
    for dirpath, _, fs in os.walk(os.path.join(ROOT, "full_waveform_inversion_amd")):
        for f in fs:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", text, re.M), f
    code = ("import sys, importlib, full_waveform_inversion_amd as p\n"
            "import os\n"
            "[importlib.import_module(p.__name__ + '.' + f[:-3]) for f in sorted(os.listdir(p.__path__[0]))"
            " if f.endswith('.py') and f != '__init__.py']\n"
            "bad = [m for m in sys.modules if m == 'oracle' or m.startswith('oracle.') or m == 'torch']\n"
            "assert not bad, bad\n")
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)


def test_shot_parallel_world2_matches_serial(tmp_path):
    """Two processes, 5 shots split 3 + 2, gradient and misfit summed over the control plane == serial."""
    out = str(tmp_path / "res")
    for rc, o in _spawn_ranks("_dist_worker.py", [out], 2):
        assert rc == 0, o
    J, g = _serial_reference()
    res = [np.load(out + ".rank%d.npz" % r) for r in range(2)]
    assert list(res[0]["mine"]) == [0, 2, 4] and list(res[1]["mine"]) == [1, 3]
    for r in res:  # every rank holds the full sum
        assert abs(float(r["J"]) - J) <= 1e-12 * J
        assert np.linalg.norm(r["g"] - g) <= 1e-12 * np.linalg.norm(g)


@pytest.mark.parametrize("first_step,branches", [(10.0, False), (3000.0, True)])
def test_lbfgs_outer_loop_world2_stays_in_lock_step(tmp_path, first_step, branches):
    """configs[4]'s pattern on two ranks: 5 shots split 3 + 2, three L-BFGS iterations, every rank running the optimiser
    on the summed gradient.  The ranks' iterates, misfits and line-search logs must be BIT-identical to each other (a
    line-search decision taken differently on one rank would deadlock or diverge the job) and equal the serial run;
    `first_step=3000` makes the first trial step fail the Armijo test, so the search branches (backtracking)."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _lbfgs_dist_worker as lw
    out = str(tmp_path / "lb")
    for rc, o in _spawn_ranks("_lbfgs_dist_worker.py", [out, repr(first_step)], 2):
        assert rc == 0, o
    res = [np.load(out + ".rank%d.npz" % r) for r in range(2)]
    logs = [json.loads(bytes(r["log"]).decode()) for r in res]
    assert np.array_equal(res[0]["x"], res[1]["x"]) and float(res[0]["f"]) == float(res[1]["f"]) and logs[0] == logs[1]
    x, f, log = lw.run(first_step)
    assert np.linalg.norm(res[0]["x"] - x) <= 1e-12 * np.linalg.norm(x) and abs(float(res[0]["f"]) - f) <= 1e-10 * abs(f)
    assert [(e["iter"], e["evals"]) for e in logs[0]] == [(e["iter"], e["evals"]) for e in log]
    assert all(abs(a.get("step", 0) - b.get("step", 0)) <= 1e-9 * abs(b.get("step", 0)) for a, b in zip(logs[0], log))
    assert len(log) == 4 and log[-1]["f"] < 0.5 * log[0]["f"]
    assert (log[1]["evals"] - log[0]["evals"] > 1 and log[1]["step"] < 1.0) == branches  # the Armijo test failed first
    # one writer: rank 0's state file is the finished run
    from full_waveform_inversion_amd.lbfgs import load_state
    st = load_state(out + ".state.npz")
    assert st["it"] == 3 and np.array_equal(st["x"], res[0]["x"])


def _run_mc_world2(tmp_path, backend):
    out = str(tmp_path / "mc")
    for rc, o in _spawn_ranks("_mc_dist_worker.py", [out, backend], 2):
        assert rc == 0, o
    return [np.load(out + ".rank%d.npz" % r) for r in range(2)]


def test_monte_carlo_sample_sharding_world2(tmp_path):
    """Sample-index sharding of the Monte Carlo loop (the reference's process fan-out, :816-848): two
    ranks' blocks together are the single-rank run, and the one exchanged scalar normalises the posterior
    over all samples.  The device call is replaced by the oracle here (no GPU in this suite)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _mc_dist_worker as w
    from full_waveform_inversion_amd import source_inversion as si
    assert [si.partition_samples(10, r, 4) for r in range(4)] == [(0, 3), (3, 3), (6, 2), (8, 2)]
    res = _run_mc_world2(tmp_path, "oracle")
    d, G, N, typ = w.problem()
    M, frac, sim, like, post = w.oracle_invert(d, G, N, typ, 9, 0, 1.5, "VR", False, False)
    assert (int(res[0]["first"]), int(res[1]["first"])) == (0, 501)
    assert res[0]["M"].shape == (10, 501) and res[1]["M"].shape == (10, 500)
    assert np.array_equal(np.hstack([r["M"] for r in res]), np.vstack((M, frac)))
    assert np.array_equal(np.concatenate([r["like"] for r in res]), like)
    allpost = np.concatenate([r["post"] for r in res])
    assert abs(allpost.sum() - 1.0) < 1e-12 and np.allclose(allpost, post, rtol=1e-12, atol=0)


def test_lbfgs_quadratic_and_rosenbrock():
    rng = np.random.default_rng(0)
    A = rng.standard_normal((20, 20))
    A = A @ A.T + 20 * np.eye(20)
    b = rng.standard_normal(20)
    x, f, log = lbfgs(lambda x: (0.5 * x @ A @ x - b @ x, A @ x - b), np.zeros(20), maxiter=40, history=8,
                      first_step=0.1)
    assert np.linalg.norm(A @ x - b) < 1e-6 * np.linalg.norm(b)
    assert all(log[i + 1]["f"] <= log[i]["f"] for i in range(len(log) - 1))

    def rosen(x):
        f = 100 * (x[1] - x[0] ** 2) ** 2 + (1 - x[0]) ** 2
        g = np.array([-400 * x[0] * (x[1] - x[0] ** 2) - 2 * (1 - x[0]), 200 * (x[1] - x[0] ** 2)])
        return f, g

    x, f, _ = lbfgs(rosen, np.array([-1.2, 1.0]), maxiter=200, history=10, first_step=0.1, max_ls=30)
    assert f < 1e-8 and np.allclose(x, 1.0, atol=1e-3)


def test_lbfgs_state_save_and_resume_reproduce_the_uninterrupted_run(tmp_path):
    """SURVEY s.5 "checkpoint / resume": the optimiser state is written after every iteration; a run resumed from the
    file of iteration 3 reproduces the uninterrupted 7-iteration run BIT FOR BIT (model, misfit, log) and evaluates
    nothing twice; the file survives being resumed from repeatedly, and a dict works like a path."""
    import shutil
    from full_waveform_inversion_amd.lbfgs import load_state

    def rosen(x):  # extended Rosenbrock, 12 unknowns
        f = float(np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1.0 - x[:-1]) ** 2))
        g = np.zeros_like(x)
        g[:-1] = -400.0 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2.0 * (1.0 - x[:-1])
        g[1:] += 200.0 * (x[1:] - x[:-1] ** 2)
        return f, g

    x0 = np.linspace(-1.2, 1.0, 12)
    calls = {"n": 0}

    def counted(x):
        calls["n"] += 1
        return rosen(x)

    ck, ck3 = str(tmp_path / "state.npz"), str(tmp_path / "state_it3.npz")
    x_ref, f_ref, log_ref = lbfgs(counted, x0, maxiter=7, history=3, first_step=0.1, max_ls=20, bounds=(-2.0, 2.0),
                                  checkpoint=ck, callback=lambda it, *_: it == 3 and shutil.copy(ck, ck3))
    total = calls["n"]
    assert load_state(ck)["it"] == 7 and load_state(ck3)["it"] == 3 and len(load_state(ck3)["S"]) == 3
    calls["n"] = 0
    x2, f2, log2 = lbfgs(counted, None, maxiter=7, history=3, first_step=0.1, max_ls=20, bounds=(-2.0, 2.0), resume=ck3)
    assert np.array_equal(x2, x_ref) and f2 == f_ref and log2 == log_ref
    assert calls["n"] == total - log_ref[3]["evals"]          # only the evaluations after iteration 3
    x3, f3, log3 = lbfgs(counted, None, maxiter=7, history=3, first_step=0.1, max_ls=20, bounds=(-2.0, 2.0),
                         resume=load_state(ck3))
    assert np.array_equal(x3, x_ref) and log3 == log_ref
    # resuming a finished run does nothing
    x4, f4, log4 = lbfgs(counted, None, maxiter=7, history=3, bounds=(-2.0, 2.0), resume=ck)
    assert np.array_equal(x4, x_ref) and log4 == log_ref
    # the continuation must share the writer's settings -- other history / bounds / shapes are errors
    assert load_state(ck3)["history"] == 3 and load_state(ck3)["bounds"] == [-2.0, 2.0]
    with pytest.raises(ValueError, match="history"):
        lbfgs(counted, None, maxiter=7, history=5, bounds=(-2.0, 2.0), resume=ck3)
    with pytest.raises(ValueError, match="bounds"):
        lbfgs(counted, None, maxiter=7, history=3, resume=ck3)
    with pytest.raises(ValueError, match="shape"):
        lbfgs(counted, np.zeros(11), maxiter=7, history=3, bounds=(-2.0, 2.0), resume=ck3)
    st = load_state(ck3)
    st["S"] = st["S"] + [st["S"][-1][:5]]
    with pytest.raises(ValueError):
        lbfgs(counted, None, maxiter=7, history=3, bounds=(-2.0, 2.0), resume=st)
    # a state written before the settings were recorded (no "history" / "bounds" keys) holding more pairs than this
    # run's history: trimmed to the newest, oldest first
    st = {k: v for k, v in load_state(ck3).items() if k not in ("history", "bounds")}
    calls["n"] = 0
    lbfgs(counted, None, maxiter=4, history=2, first_step=0.1, max_ls=20, bounds=(-2.0, 2.0), resume=st)
    assert calls["n"] >= 1


def test_lbfgs_checkpoints_a_failed_line_search_and_a_resume_does_not_repeat_it(tmp_path):
    """a run that ends with "line search failed" writes that terminal log entry too; resuming from the
    file returns at once instead of repeating max_ls misfit evaluations (each one a sweep over all shots)."""
    from full_waveform_inversion_amd.lbfgs import load_state
    calls = {"n": 0}

    def bumpy(x):  # the gradient lies about the slope: no step along -g ever satisfies the Armijo test
        calls["n"] += 1
        return float(x @ x), -2.0 * x

    ck = str(tmp_path / "s.npz")
    x, f, log = lbfgs(bumpy, np.ones(4), maxiter=3, first_step=0.5, max_ls=4, checkpoint=ck)
    assert log[-1].get("note") == "line search failed" and calls["n"] == 1 + 4
    st = load_state(ck)
    assert st["log"] == log and st["it"] == 0 and np.array_equal(st["x"], x)
    calls["n"] = 0
    x2, f2, log2 = lbfgs(bumpy, None, maxiter=3, first_step=0.5, max_ls=4, resume=ck)
    assert calls["n"] == 0 and log2 == log and np.array_equal(x2, x) and f2 == f


def test_lbfgs_recovers_from_a_far_too_long_first_step_in_few_evaluations():
    """A first step 100x too long: interpolating the parabola through f(0), f'(0) and the failed trial finds the
    scale in two or three misfit evaluations (each one a sweep over all shots) where halving needs seven."""
    rng = np.random.default_rng(3)
    A = np.diag(1.0 + 4.0 * rng.random(30))
    b = rng.standard_normal(30)
    g0 = np.abs(b).max()
    exact = g0 * (b @ b) / (b @ A @ b)      # first_step that lands on the minimiser along -g
    x, f, log = lbfgs(lambda x: (0.5 * x @ A @ x - b @ x, A @ x - b), np.zeros(30), maxiter=3, history=4,
                      first_step=100.0 * exact)
    assert log[1]["evals"] - log[0]["evals"] <= 3 and log[1]["f"] < log[0]["f"]
    assert 0.2 / 100.0 < log[1]["step"] < 2.0 / 100.0


def test_lbfgs_expands_a_too_short_first_step():
    """The first step of an FWI run is scaled by hand (`first_step`); when it is far too short the slope along
    the direction is still steep at the trial point (weak Wolfe curvature fails) and the search doubles the step
    instead of accepting it -- and non-finite misfits are an error, not "converged"."""
    def quad(x):
        return 0.5 * float(x @ x), x.copy()
    x0 = np.full(6, 100.0)
    _, f_w, log_w = lbfgs(quad, x0, maxiter=1, first_step=1e-3, max_ls=30)
    _, f_a, log_a = lbfgs(quad, x0, maxiter=1, first_step=1e-3, max_ls=30, c2=None)
    assert log_w[1]["step"] > 1000 * log_a[1]["step"] and f_w < 0.75 * f_a  # (c2 = 0.9: accepted once the slope has dropped 10 %)
    with pytest.raises(FloatingPointError):
        lbfgs(lambda x: (float("nan"), x), x0, maxiter=2)
    with pytest.raises(ValueError):
        lbfgs(quad, x0, history=0)


def test_lbfgs_respects_bounds():
    x, f, _ = lbfgs(lambda x: (float(np.sum((x - 3.0) ** 2)), 2 * (x - 3.0)), np.zeros(4), maxiter=20,
                    first_step=1.0, bounds=(-1.0, 2.0))
    assert np.allclose(x, 2.0)


def test_small_inversion_reduces_misfit_with_oracle_engine():
    """The whole outer loop (shots -> gradient -> L-BFGS) on a toy 2-D problem, CPU oracle engine."""
    w = workloads.cfg3(0.0625, nshots=3)
    wav = w.wavelet(np.float64)
    shots = [sh.Shot(w.src_idx[i:i + 1], wav, w.rec_idx) for i in range(3)]
    e = OracleEngine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml)
    sh.model_data(e, w.c, shots)
    _, _, log = lbfgs(lambda m: sh.misfit_and_gradient(e, m, shots), w.c_init, maxiter=4, history=4,
                      first_step=30.0, bounds=(1000.0, 4000.0))
    assert log[-1]["f"] < 0.5 * log[0]["f"]


def test_engine_pool_matches_single_engine():
    """Three engines sharing the shots (threads) give the single-engine misfit and gradient."""
    w = workloads.cfg3(0.0625, nshots=5)
    wav = w.wavelet(np.float64)

    def mk():
        return OracleEngine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml, sigma_max=800.0)

    shots1 = [sh.Shot(w.src_idx[i:i + 1], wav, w.rec_idx) for i in range(5)]
    shots3 = [sh.Shot(w.src_idx[i:i + 1], wav, w.rec_idx) for i in range(5)]
    e = mk()
    sh.model_data(e, w.c, shots1)
    J1, g1 = sh.misfit_and_gradient(e, w.c_init, shots1)
    with sh.EnginePool(mk, 3) as pool:
        sh.model_data(pool, w.c, shots3)
        assert all(np.array_equal(a.d_obs, b.d_obs) for a, b in zip(shots1, shots3))
        J3, g3 = sh.misfit_and_gradient(pool, w.c_init, shots3)
    assert abs(J3 - J1) <= 1e-13 * J1 and np.linalg.norm(g3 - g1) <= 1e-13 * np.linalg.norm(g1)


def test_engine_pool_propagates_worker_errors():
    w = workloads.cfg3(0.0625, nshots=3)
    shots = [sh.Shot(w.src_idx[i:i + 1], w.wavelet(np.float64), w.rec_idx) for i in range(3)]  # no d_obs
    with sh.EnginePool(lambda: OracleEngine(w.shape, w.h, w.dt, w.nt, order=w.order, npml=w.npml), 2) as pool:
        with pytest.raises(ValueError):
            sh.misfit_and_gradient(pool, w.c_init, shots)
