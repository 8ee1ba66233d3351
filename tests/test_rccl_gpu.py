"""The multi-GPU path of bench.py on the GPU box (one rank): torch.distributed.run -> RCCL process group ->
sharded solve -> the summary all-reduce over backend "nccl" (= RCCL). The driver's N>1 runs launch bench.py exactly
like this with more ranks; here the collective path is executed with world_size 1 so that it is not first run on
the 8-GPU node. Also: two handles on two devices from one process, where the box has more than one GPU."""
from __future__ import annotations

import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
from conftest import ROOT, golden, rel_err

pytestmark = pytest.mark.gpu


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_under_torchrun_uses_rccl_and_reduces_the_summary(pkg):
    batch, iters = 512, 200
    env = dict(os.environ)
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)  # bench.py sets what its RCCL path needs itself (rccl_environment)
    env.pop("TINYMPC_LAYOUT", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--batch-per-gpu", str(batch), "--iters", str(iters), "--no-cpu-baseline", "--no-single", "--no-config5"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)  # a child process: nothing is exec'ed over this one
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["config"]["global_batch"] == batch
    s = out["summary"]  # produced by batch.allreduce_summary over the RCCL group
    assert s["instances"] == batch and s["total_iterations"] == batch * iters and s["converged"] == 0
    assert out["process_group"] == {"backend": "nccl", "world_size": 1, "ranks_per_device": 1}
    assert len(out["per_rank"]) == 1 and out["per_rank"][0]["instances"] == batch and out["per_rank"][0]["kernel_ms_avg"] > 0
    # the numbers in the summary are those of the solve: residual maxima against the golden prefix
    g = golden("quadrotor_batch64")
    assert out["parity_check"] is not None and out["parity_check"]["ok"]
    assert s["max_primal_residual"] >= float(np.max(g["residuals"][[0, 2]])) * (1 - 1e-6)


def test_bench_two_ranks_over_rccl(pkg):
    """`bench.py --gpus 2` exactly as the driver launches it (torchrun, one rank per GPU) -- where the box has two GPUs."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible")
    batch, iters = 512, 200
    env = dict(os.environ)
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
    env.pop("TINYMPC_LAYOUT", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch-per-gpu", str(batch), "--iters", str(iters), "--no-cpu-baseline", "--no-single", "--no-config5"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 2 * batch and out["scaling"] == "weak"
    assert out["summary"]["instances"] == 2 * batch and out["summary"]["total_iterations"] == 2 * batch * iters
    assert [p["first_instance"] for p in out["per_rank"]] == [0, batch] and all(p["instances"] == batch for p in out["per_rank"])
    assert out["process_group"] == {"backend": "nccl", "world_size": 2, "ranks_per_device": 1}


def test_two_handles_on_two_devices(pkg):
    """Single process, one handle per device (`device=`): what a caller does that drives several GPUs itself."""
    if pkg.device_count() < 2:
        pytest.skip("one GPU visible")
    P = pkg.problems
    prob = P.quadrotor(50)
    x0s = P.quadrotor_batch_x0(128)
    sols = []
    for dev in (0, 1):
        s = pkg.TinyMPC()
        s.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=64, device=dev, rho=prob.rho, abs_pri_tol=0.0, abs_dua_tol=0.0,
                max_iter=50)
        s.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
        s.set_x0_batch(x0s[:, dev * 64:(dev + 1) * 64])
        s.solve_async()
        sols.append(s)
    one = pkg.TinyMPC()
    one.setup(prob.A, prob.B, prob.Q, prob.R, prob.N, batch=128, device=0, rho=prob.rho, abs_pri_tol=0.0, abs_dua_tol=0.0, max_iter=50)
    one.set_bound_constraints(prob.x_min, prob.x_max, prob.u_min, prob.u_max)
    one.set_x0_batch(x0s)
    one.solve()
    ref = one.get_solution_batch()
    for dev, s in enumerate(sols):
        s.synchronize()
        got = s.get_solution_batch()
        np.testing.assert_array_equal(got["controls"], ref["controls"][:, :, dev * 64:(dev + 1) * 64])
        s.reset()
    one.reset()
