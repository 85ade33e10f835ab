"""The two Python examples run to completion on a GPU and report success through their exit code."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,args", [("fwi_2d_demo.py", ["--scale", "0.125", "--shots", "4", "--iters", "4"]),
                                         ("source_inversion_demo.py", ["--samples", "200000"])])
def test_example_runs(gpu, script, args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script)] + args, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, (p.stdout.decode()[-1500:], p.stderr.decode()[-1500:])
