"""bench.py's N > 1 path with REAL HIP solves in every rank, on the one GPU of the box.

The driver launches `bench.py --gpus N` under torch.distributed.run with one rank per GPU over RCCL. A one-GPU box cannot do
that (RCCL refuses two ranks on one device), so the same code path -- job_shard -> per-rank x0 offsets -> barrier-bracketed timed
steps -> MAX all-reduce -> per_rank all-gather -> summary all-reduce -- is run here with `--share-device --dist-backend gloo`:
every rank opens device 0 and solves its own shard on it, the collectives go through host memory. What the ranks compute must
not depend on the sharding, bit for bit: each rank's first controls are fingerprinted in the JSON line (`per_rank`) and compared
with ONE unsharded solve of the whole batch in this process. (the semantics are those of independent solvers,
/root/reference/src/codegen_src/tinympc/admm.cpp:109-207.)

Rank counts: the GPU pool allows six processes on a card at once and this pytest process is one of them, so the suite runs 2 and 4
ranks -- the torchrun launcher holds the device open as well, so five ranks is the most a clean parent can start (`python bench.py --gpus 5
--share-device --dist-backend gloo`:
profiles/r04_multirank_5.json; six ranks were killed by the pool's process guard); the split arithmetic for 8 ranks is covered
on the CPU (tests/test_distributed_cpu.py::test_job_shard_weak_and_strong)."""
from __future__ import annotations

import hashlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
from conftest import ROOT, golden

pytestmark = pytest.mark.gpu

ITERS = 200


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(world: int, extra: list[str], timeout: int = 900):
    env = dict(os.environ)
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)  # bench.py sets what it needs itself (rccl_environment)
    env.pop("TINYMPC_LAYOUT", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
           "--iters", str(ITERS), "--no-cpu-baseline", "--no-single", "--no-config5"] + extra
    # fresh child processes (torchrun starts one per rank); nothing is exec'ed over a process that has touched the GPU
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


def _line(r) -> dict:
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


@pytest.fixture(scope="module")
def unsharded(pkg):
    """First controls and iteration counts of ONE cold-started solve of `total` instances in this process, cached per total."""
    cache = {}

    def run(total: int):
        if total not in cache:
            P = pkg.problems
            prob = P.quadrotor(50)
            s = pkg.TinyMPC()
            s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=total, rho=prob.rho, abs_pri_tol=0.0, abs_dua_tol=0.0,
                    max_iter=ITERS, check_termination=1)
            s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
            s.set_x0_batch(np.asfortranarray(P.quadrotor_batch_x0(total)))
            s.reset_workspace()
            s.solve()
            cache[total] = (np.ascontiguousarray(s.get_first_controls_batch().T), s.get_stats_batch()["iter"].copy())
            s.reset()
        return cache[total]

    return run


def _check_line(pkg, out, world, total, scaling, unsharded):
    assert out["n_gpus"] == world and out["scaling"] == scaling and out["config"]["global_batch"] == total
    assert out["process_group"]["backend"] == "gloo" and out["process_group"]["world_size"] == world
    assert "share device" in str(out["process_group"]["ranks_per_device"])
    pr = out["per_rank"]
    assert [p["rank"] for p in pr] == list(range(world)) and all(p["device"] == 0 for p in pr)
    shards = [pkg.batch.shard_range(total, r, world) for r in range(world)]
    assert [(p["first_instance"], p["instances"]) for p in pr] == shards  # contiguous, in rank order
    assert all(p["kernel_ms_avg"] > 0 and p["elapsed_s"] > 0 for p in pr)
    s = out["summary"]
    assert s["instances"] == total and s["total_iterations"] == total * ITERS and s["converged"] == 0
    assert sum(p["total_iterations"] for p in pr) == total * ITERS
    # the value is the whole job's: all ranks' instance-iterations over the MAX of the ranks' elapsed times
    assert out["value"] == pytest.approx(total * ITERS * out["steps"] / max(p["elapsed_s"] for p in pr), rel=1e-9)
    assert out["value_as_asked"] > 0 and out["legs"]["value_as_asked"] == out["value_as_asked"]
    # rank 0's 64-instance prefix against the reference core's own output
    assert out["parity_check"] is not None and out["parity_check"]["ok"] and out["parity_check"]["iterations_match"]
    # sharding changes nothing, bit for bit: every rank's first controls == the unsharded solve's slice
    u0, its = unsharded(total)
    assert int(its.sum()) == total * ITERS
    for p, (first, count) in zip(pr, shards):
        mine = u0[first:first + count]
        assert p["first_controls_sha256"] == hashlib.sha256(mine.tobytes()).hexdigest(), "rank %d" % p["rank"]
        np.testing.assert_array_equal(np.array(p["first_controls_prefix"]), mine[:4])
    g = golden("quadrotor_batch64")
    np.testing.assert_allclose(np.array(pr[0]["first_controls_prefix"]), g["sol_u"][:, 0, :4].T, rtol=1e-9, atol=1e-12)
    assert "skipped" in out["legs_run_on"]


@pytest.mark.parametrize("world", [2, 4])
def test_weak_scaling_ranks_share_the_device(pkg, unsharded, world):
    batch = 1024  # (above 768 instances every shard and the unsharded batch run the same kernel, layout D)
    out = _line(_launch(world, ["--share-device", "--dist-backend", "gloo", "--batch-per-gpu", str(batch)]))
    _check_line(pkg, out, world, world * batch, "weak", unsharded)
    assert out["launch"]["layout"] == "D"


def test_strong_scaling_with_ragged_shards(pkg, unsharded):
    world, total = 4, 4099  # 1025, 1025, 1025, 1024
    out = _line(_launch(world, ["--share-device", "--dist-backend", "gloo", "--global-batch", str(total)]))
    _check_line(pkg, out, world, total, "strong", unsharded)
    assert [p["instances"] for p in out["per_rank"]] == [1025, 1025, 1025, 1024]


def test_devices_list_is_the_same_as_share_device(pkg, unsharded):
    out = _line(_launch(2, ["--devices", "0,0", "--dist-backend", "gloo", "--batch-per-gpu", "1024"]))
    _check_line(pkg, out, 2, 2048, "weak", unsharded)


def test_a_rank_that_fails_its_setup_fails_the_job(pkg):
    """Rank 1 is given a device that does not exist: it exits non-zero before the rendezvous, torchrun ends the other rank, and the
    job's exit code is non-zero with no JSON line -- a driver cannot mistake it for a measurement."""
    r = _launch(2, ["--devices", "0,99", "--dist-backend", "gloo", "--batch-per-gpu", "1024"], timeout=600)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "device 99 does not exist" in r.stderr


def test_shared_device_over_rccl_is_refused_with_a_message(pkg):
    r = _launch(2, ["--share-device", "--batch-per-gpu", "1024"], timeout=600)
    assert r.returncode != 0 and "need --dist-backend gloo" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
