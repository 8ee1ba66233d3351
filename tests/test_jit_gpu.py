"""The run-time specialiser (tinympc_jit.hip) on the GPU box: the library is SELF-CONTAINED (kernel sources embedded at build
time -- a copied libtinympc_hip.so alone specialises new shapes) and HONEST (a specialisation whose code object spills or needs
more registers than its plan is refused, says why, and the handle runs on a generic kernel). Each case runs in a child process:
the library path, the cache directory and the specialiser's switches are process-wide environment."""
from __future__ import annotations

import json
import os
import shutil
import subprocess
import sys

import pytest
from conftest import ROOT

pytestmark = pytest.mark.gpu

CHILD = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "oracle"))
import __graft_entry__ as ge
import pyoracle as O
pkg = ge.load_package()
P = pkg.problems
prob = P.quadrotor(%(N)d)
batch = 2048
s = pkg.TinyMPC()
s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=batch, rho=prob.rho, abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=30)
s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
x0s = P.quadrotor_batch_x0(batch)
s.set_x0_batch(x0s)
s.solve()
sol = s.get_solution_batch()
o = O.OraclePort(prob).load_problem(prob, dict(abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=30))
o.set_x0(x0s[:, 777]); o.solve()
err = float(np.max(np.abs(sol["controls"][:, :, 777] - o.solution()[1])) / np.max(np.abs(o.solution()[1])))
print(json.dumps(dict(layout=s.launch_info()["layout"], jit=s.jit_info(), lib=pkg._lib.LIB_PATH, err=err)))
s.reset()
"""


def run_child(env_extra, N=40):
    env = dict(os.environ)
    env.pop("TINYMPC_LAYOUT", None)
    env.pop("TINYMPC_HIP_SRC", None)
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, N=N)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]), r.stderr


def test_a_copied_library_specialises_without_its_source_tree(tmp_path):
    """Only libtinympc_hip.so travels to an empty directory (no csrc/, no include/ next to it), with an empty code-object cache:
    quadrotor N=40 x 2,048 -- a shape that is not compiled in -- must still get its layout-D kernel, freshly compiled."""
    lib = os.path.join(ROOT, "tinympc-matlab_amd", "libtinympc_hip.so")
    alone = tmp_path / "elsewhere"
    alone.mkdir()
    shutil.copy(lib, alone / "libtinympc_hip.so")
    assert sorted(os.listdir(alone)) == ["libtinympc_hip.so"]
    out, _ = run_child({"TINYMPC_HIP_LIBRARY": str(alone / "libtinympc_hip.so"), "TINYMPC_JIT_CACHE": str(tmp_path / "cache")})
    assert out["lib"] == str(alone / "libtinympc_hip.so")
    assert out["layout"] == "D", out
    assert out["jit"].startswith("compiled ") and "scratch=0" in out["jit"], out
    assert out["err"] < 1e-9
    # the same shape again: now from the disk cache (0700 directory, checksummed image)
    out2, _ = run_child({"TINYMPC_HIP_LIBRARY": str(alone / "libtinympc_hip.so"), "TINYMPC_JIT_CACHE": str(tmp_path / "cache")})
    assert out2["layout"] == "D" and out2["jit"].startswith("disk-cache "), out2
    mode = os.stat(tmp_path / "cache").st_mode & 0o777
    assert mode == 0o700, oct(mode)
    # a cache directory that others may write to is not trusted: compiled again, nothing loaded from it
    os.chmod(tmp_path / "cache", 0o777)
    out3, _ = run_child({"TINYMPC_HIP_LIBRARY": str(alone / "libtinympc_hip.so"), "TINYMPC_JIT_CACHE": str(tmp_path / "cache")})
    assert out3["layout"] == "D" and out3["jit"].startswith("compiled "), out3
    # a corrupted image is detected by its checksum and replaced
    os.chmod(tmp_path / "cache", 0o700)
    images = [f for f in os.listdir(tmp_path / "cache") if f.endswith(".hsaco")]
    assert images
    for f in images:
        path = tmp_path / "cache" / f
        blob = bytearray(path.read_bytes())
        blob[len(blob) // 2] ^= 0xFF
        path.write_bytes(bytes(blob))
    out4, _ = run_child({"TINYMPC_HIP_LIBRARY": str(alone / "libtinympc_hip.so"), "TINYMPC_JIT_CACHE": str(tmp_path / "cache")})
    assert out4["layout"] == "D" and out4["jit"].startswith("compiled ") and out4["err"] < 1e-9, out4


def test_a_specialisation_beyond_its_register_plan_is_refused_loudly(tmp_path):
    """TINYMPC_JIT_REG_LIMIT lowers the budget the compiled code object is checked against: the kernel is built, its metadata read
    back, and the launch REFUSED with the numbers in the reason -- on stderr once and in tinympc_get_jit_info --; the handle then
    runs on a generic kernel and still solves."""
    out, err = run_child({"TINYMPC_JIT_CACHE": str(tmp_path / "cache"), "TINYMPC_JIT_REG_LIMIT": "64"})
    assert out["layout"] != "D", out
    assert out["jit"].startswith("refused(") and "registers, the plan allows 64" in out["jit"], out
    assert "run-time specialisation refused" in err and "the plan allows 64" in err
    assert out["err"] < 1e-9
    quiet, err2 = run_child({"TINYMPC_JIT_CACHE": str(tmp_path / "cache"), "TINYMPC_JIT_REG_LIMIT": "64", "TINYMPC_JIT_QUIET": "1"})
    assert quiet["jit"].startswith("refused(") and "run-time specialisation refused" not in err2
    off, _ = run_child({"TINYMPC_JIT": "0"})
    assert off["layout"] != "D" and off["jit"] == "refused(TINYMPC_JIT=0)", off
