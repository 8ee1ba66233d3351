"""Multi-process CPU test of the batched mode (gloo, world_size 2): contiguous sharding, zero data-path
exchange, one summary all-reduce -- the N>1 path of bench.py by construction."""
from __future__ import annotations

import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
from conftest import ROOT

import pyoracle as O


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("total,mode", [(10, "weak"), (7, "weak"), (9, "strong")])
def test_two_rank_sharded_batch_matches_unsharded(pkg, tmp_path, total, mode):
    out = tmp_path / "result.json"
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_dist_worker.py"), str(out), str(total), mode]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(out.read_text())
    # unsharded reference computation
    P = pkg.problems
    prob = P.quadrotor(20)
    settings = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=120, check_termination=1)
    orc = O.OraclePort(prob).load_problem(prob, settings)
    sx, su, iters, status, res = orc.solve_batch(P.quadrotor_batch_x0(total))
    s = got["summary"]
    assert got["world"] == 2 and got["shard"] == list(pkg.batch.shard_range(total, 0, 2))
    assert s["instances"] == total
    assert s["converged"] == int(np.sum(status == 1))
    assert s["total_iterations"] == int(np.sum(iters))
    assert s["max_primal_residual"] == pytest.approx(float(np.max(res[[0, 2]])), rel=1e-12)
    assert s["max_dual_residual"] == pytest.approx(float(np.max(res[[1, 3]])), rel=1e-12)
    np.testing.assert_array_equal(np.array(got["u0"]), su[:, 0, :].T)  # sharding changes nothing, bit for bit


def test_summary_without_process_group(pkg):
    B = pkg.batch
    s = B.allreduce_summary(B.local_summary(np.array([3, 5]), np.array([1, 11]), np.array([[1e-3, 2e-3], [3e-3, 1e-3], [5e-4, 1e-4], [2e-3, 9e-3]])))
    assert s == dict(instances=2, converged=1, total_iterations=8, max_primal_residual=2e-3, max_dual_residual=9e-3)


def test_job_shard_weak_and_strong(pkg):
    """bench.py's split: --batch-per-gpu fixes the shard (weak), --global-batch the total (strong, ragged where it does not divide)."""
    B = pkg.batch
    for world in (1, 2, 3, 8):
        shards = [B.job_shard(r, world, batch_per_gpu=8192) for r in range(world)]
        assert all(s[0] == 8192 * world and s[2] == 8192 and s[3] == "weak" for s in shards)
        assert [s[1] for s in shards] == [8192 * r for r in range(world)]
        for total in (65536, 1001, world):
            shards = [B.job_shard(r, world, batch_per_gpu=8192, global_batch=total) for r in range(world)]
            assert all(s[0] == total and s[3] == "strong" for s in shards)
            assert sum(s[2] for s in shards) == total and max(s[2] for s in shards) - min(s[2] for s in shards) <= 1
            assert [s[1] for s in shards] == [sum(t[2] for t in shards[:r]) for r in range(world)]  # contiguous, in rank order
    with pytest.raises(ValueError):
        B.job_shard(0, 8, global_batch=5)
