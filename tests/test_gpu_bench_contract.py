"""The bench line's contract (driver: one JSON line with metric / value / roofline / cpu_baseline ...), checked on a
small grid so that it runs in seconds: `bench.py --leg headline` and one extra leg."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_headline_line_has_the_contract_fields(gpu):
    d = _run("--leg", "headline", "--grid", "64", "--nt", "40", "--steps", "2", "--warmup", "1")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["unit"] == "Gpts/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["dtype"] == "f32"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["rccl_ranks"] is None
    r = d["roofline"]
    assert r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["bound"] == "infinity-cache-resident"  # a 64^3 working set certainly is
    # value = updates / wall time, consistent with ms_per_step
    assert abs(d["value"] - 64 ** 3 * 40 / (d["ms_per_step"] * 1e-3) / 1e9) < 0.02 * d["value"]


def test_an_extra_leg_alone_reports_its_roofline(gpu):
    d = _run("--leg", "cfg2", "--leg-nt", "200")
    leg = d["legs"]["cfg2"]
    assert leg["kernel"] == "step2d_fused" and leg["Gpts_per_s"] > 50 and leg["us_per_time_step"] > 0
    assert leg["algorithmic_bytes_per_launch"] == 4 * 16 * 1024 ** 2


def test_more_ranks_than_launched_is_an_error_not_a_silent_single_gpu_run(gpu):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=120, env={k: v for k, v in os.environ.items()
                                                                  if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert p.returncode != 0 and b"torch.distributed.run" in p.stderr


def test_the_multi_rank_path_as_the_driver_launches_it_with_one_rank(gpu):
    """`python -m torch.distributed.run --nproc-per-node 1 ... bench.py` with FWI_BENCH_FORCE_EXCHANGE=1: the launcher,
    the stdlib rendezvous on MASTER_PORT + 1.., ncclCommInitRank, the warm-up and the timed all-reduce of the gradient
    accumulator and the max-over-ranks reduction all run -- everything of the N = 2, 4, 8 runs except a second GPU."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, FWI_BENCH_FORCE_EXCHANGE="1", FWI_BENCH_FORCE_2D_SCALING="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--leg", "headline", "--grid", "64", "--nt", "40", "--steps", "2", "--warmup",
                        "1", "--leg-nt", "200"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    # ... and nothing else on stdout: RCCL's version banner at communicator init is sent to stderr (bench.py, exchange)
    assert [ln for ln in p.stdout.decode().splitlines() if ln.strip()] == lines, p.stdout.decode()[:600]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["rccl_ranks"] == 1 and "rccl allreduce" in d["config"]["exchange"]
    assert d["value"] > 0
    # the 2-D configuration rides along in N > 1 runs (control-plane barriers only); optional, so errors are recorded
    leg = d["legs"]["cfg2"]
    assert "error" not in leg, leg
    assert leg["n_gpus"] == 1 and leg["kernel"] == "step2d_fused" and leg["Gpts_per_s"] > 50


def test_the_gradient_leg_times_the_real_exchange_pattern_and_checks_the_sum(gpu):
    """forward + store, adjoint + imaging per shot, then the ONE all-reduce of the accumulator --
    rehearsed with one rank through the driver's launch line; the line carries the check g.g(after) = N^2 g.g(before)."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, FWI_BENCH_FORCE_EXCHANGE="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--leg", "gradient", "--leg-nt", "500"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600, env=env)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    d = json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][0])
    leg = d["legs"]["gradient"]
    assert leg["rccl_ranks"] == 1 and "rccl allreduce" in leg["exchange"] and leg["allreduce_ms"] > 0
    chk = leg["allreduce_check"]
    assert chk["ok"] and chk["expected_ratio"] == 1.0 and chk["g_dot_g_before_sum"] > 0
    assert leg["update_form"] == "standard" and leg["Gpts_per_s_both_sweeps"] > 20


def test_a_failed_all_reduce_check_is_fatal_not_a_field_in_the_line(gpu):
    """a wrong or skipped RCCL sum must not exit 0 with a headline figure.  Rehearsed with one rank: the
    check is told to expect the sum of TWO ranks (FWI_BENCH_CHECK_EXPECT_RANKS=2), g.g after / before is 1 instead of 4,
    every rank learns the outcome over RCCL and the process ends non-zero without a JSON line."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, FWI_BENCH_FORCE_EXCHANGE="1", FWI_BENCH_CHECK_EXPECT_RANKS="2")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--leg", "gradient", "--leg-nt", "100"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600, env=env)
    assert p.returncode != 0
    assert b"all-reduce check FAILED" in p.stderr and not [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]


def test_the_cpml_legs_report_the_kernels_that_carry_the_border(gpu):
    d = _run("--leg", "cfg2_cpml", "--leg-nt", "200")
    leg = d["legs"]["cfg2_cpml"]
    assert leg["kernel"] == "step2d_fused" and leg["us_per_time_step"] < 12.0  # (19.5 through the slab path)
    assert leg["algorithmic_bytes_per_launch"] == 4 * 16 * 1024 ** 2 + 16 * 2 * 2 * 40 * 1024
    d = _run("--leg", "cpml3d", "--leg-nt", "40")
    assert d["legs"]["cpml3d"]["kernel"] == "step3d_stream" and d["legs"]["cpml3d"]["us_per_time_step"] > 0
