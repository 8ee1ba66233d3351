"""Worker for tests/test_distributed_cpu.py: one rank of the batched mode on CPU (gloo).

Exactly the bookkeeping bench.py does per GPU -- shard_range -> seeded x0 shard -> solve the shard ->
local_summary -> allreduce_summary -- with the plain-C oracle standing in for the per-rank solver
(tests may use the oracle; on a GPU box the same code drives a TinyMPC handle). Rank 0 writes the
combined result as JSON to the path in argv[1]."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as ge  # noqa: E402
import pyoracle as O  # noqa: E402


def main():
    out_path, total = sys.argv[1], int(sys.argv[2])
    strong = len(sys.argv) > 3 and sys.argv[3] == "strong"  # bench.py --global-batch: the total is fixed, shards may be ragged
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    pkg = ge.load_package()
    P, B = pkg.problems, pkg.batch
    prob = P.quadrotor(20)
    settings = dict(abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=120, check_termination=1)
    if strong:
        total_, first, count, kind = B.job_shard(rank, world, batch_per_gpu=4096, global_batch=total)
        assert (total_, kind) == (total, "strong")
    else:
        first, count = B.shard_range(total, rank, world)
    x0s = P.quadrotor_batch_x0(count, offset=first)
    orc = O.OraclePort(prob).load_problem(prob, settings)
    sx, su, iters, status, res = orc.solve_batch(x0s)
    summary = B.allreduce_summary(B.local_summary(iters, status, res))
    # optional all-gather of the first controls (2 MB at 65,536 instances)
    u0 = torch.from_numpy(np.ascontiguousarray(su[:, 0, :].T))  # [count][nu]
    sizes = [B.shard_range(total, r, world)[1] for r in range(world)]
    gathered = [torch.zeros((n, prob.nu), dtype=torch.float64) for n in sizes]
    dist.all_gather(gathered, u0) if len(set(sizes)) == 1 else None
    if len(set(sizes)) != 1:  # ragged shards: gather via padded tensors
        pad = torch.zeros((max(sizes), prob.nu), dtype=torch.float64)
        pad[:count] = u0
        padded = [torch.zeros_like(pad) for _ in range(world)]
        dist.all_gather(padded, pad)
        gathered = [p_[:n] for p_, n in zip(padded, sizes)]
    if rank == 0:
        json.dump(dict(summary=summary, u0=torch.cat(gathered).numpy().tolist(), shard=[first, count], world=world),
                  open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
