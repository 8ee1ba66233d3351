"""The example scripts under examples/ run to completion on a GPU box (they are the documentation of the call
sequences a user of the reference switches to)."""
import os
import subprocess
import sys

import pytest
from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("script", ["one_solve.py", "mpc_loop.py", "rocket_landing.py", "batched_throughput.py", "large_system.py", "converging_batch.py"])
def test_example_runs(script):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script)], capture_output=True, text=True, timeout=600,
                         cwd=os.path.join(ROOT, "examples"))
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert res.stdout.strip(), "the example printed nothing"
