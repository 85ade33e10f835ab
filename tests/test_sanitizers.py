"""Host sanitizer builds (SURVEY.md s.5 "race detection / sanitizers").

The C oracle is small enough to rebuild with AddressSanitizer + UBSan and run inside the CPU suite; the shim's
sanitizer build (`make -C full_waveform_inversion_amd/csrc asan`, three minutes of hipcc) runs from
tests/run_asan.sh (non-zero exit on any failure or sanitizer report), whose last output is kept as profiles/r04_asan.log.  GPU sanitizers need xnack+ code objects,
which this pool does not offer: host code only."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import numpy as np
from oracle import fwi_oracle as fo
from oracle.c_oracle import CPropagator
for shape, order, npml, abc in (((21, 17, 19), 8, 4, "sponge"), ((30, 26), 4, 5, "cpml"), ((9, 8, 7), 2, 0, "sponge")):
    rng = np.random.default_rng(1)
    c = 1800.0 + 700.0 * rng.random(shape)
    dt = 0.6 * fo.cfl_dt(c.max(), 10.0, len(shape), order)
    src = np.array([[s // 2 for s in shape], [0] * len(shape)])          # a corner node: halo arithmetic at the edge
    rec = np.array([[s - 1 for s in shape], [1] * len(shape)])
    w = rng.standard_normal((23, 2))
    kw = dict(abc=abc, pml_alpha_max=20.0) if abc == "cpml" else {}
    p, q = CPropagator(c, 10.0, dt, order, npml, **kw), fo.Propagator(c, 10.0, dt, order, npml, **kw)
    d, d2 = p.forward(src, w, rec), q.forward(src, w, rec)
    a, a2 = p.adjoint(d), q.adjoint(d2)
    assert np.allclose(d, d2, rtol=1e-11, atol=1e-300) and np.allclose(a, a2, rtol=1e-10, atol=1e-300)
    assert np.allclose(p.gradient(), q.gradient(), rtol=1e-9, atol=1e-300)
print("asan-ok")
"""


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_c_oracle_under_address_and_ub_sanitizers():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    if not os.path.isabs(rt) or not os.path.exists(rt):
        pytest.skip("gcc has no libasan")
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0", OMP_NUM_THREADS="2", PYTHONPATH=ROOT,
               FWI_ORACLE_LIB=os.path.join(ROOT, "oracle", "libfwi_oracle_asan.so"))
    p = subprocess.run([sys.executable, "-c", SCRIPT], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0 and "asan-ok" in out and "ERROR: AddressSanitizer" not in out \
        and "runtime error" not in out, out[-3000:]


def test_the_sanitizer_script_fails_when_a_run_fails_or_reports():
    """tests/run_asan.sh used to return tee's status and let sanitizer reports scroll by.  It now exits
    non-zero on a failed build, a failed pytest run or a report in the output -- checked here on its logic (the full
    pass is three minutes of hipcc, run by hand / tools; its last log is profiles/r04_asan.log)."""
    sh = open(os.path.join(ROOT, "tests", "run_asan.sh")).read()
    assert "pipefail" in sh and "pytest-rc-shim=0" in sh and "pytest-rc-oracle=0" in sh and "exit $rc" in sh
    assert "ERROR: AddressSanitizer" in sh and "runtime error:" in sh
    assert "| tail -15" not in sh  # (the pipes that hid pytest's status)
    mk = open(os.path.join(ROOT, "full_waveform_inversion_amd", "csrc", "Makefile")).read()
    assert "-fsanitize=address,undefined" in mk and "asan:" in mk
