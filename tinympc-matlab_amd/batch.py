"""Batched mode across GPUs: one process per GPU, contiguous shards of independent MPC instances.

MPC instances share only read-only problem data: no instance reads another's
state, so a solve needs NO data-path collective. The only exchange is one tiny reduction of summary
statistics after the solve (sum of converged counts / iterations, max of residuals) -- over RCCL
(torch.distributed backend "nccl") on GPUs, over gloo in the CPU tests.
"""
from __future__ import annotations

import numpy as np


def shard_range(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous split of instance indices: returns (first, count) for `rank`. The first
    total % world ranks get one extra instance."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world: {rank}/{world}")
    base, extra = divmod(int(total), int(world))
    first = rank * base + min(rank, extra)
    count = base + (1 if rank < extra else 0)
    return first, count


def job_shard(rank: int, world: int, batch_per_gpu: int = 0, global_batch: int = 0) -> tuple[int, int, int, str]:
    """The split bench.py runs: weak scaling (`batch_per_gpu` instances on every rank) unless `global_batch` > 0 fixes
    the TOTAL (strong scaling; ragged shards when it does not divide). Returns (total, first, count, "weak"|"strong")."""
    strong = global_batch > 0
    total = int(global_batch) if strong else int(batch_per_gpu) * int(world)
    if total < world:
        raise ValueError(f"{total} instances cannot be split over {world} ranks")
    first, count = shard_range(total, rank, world)
    return total, first, count, "strong" if strong else "weak"


def local_summary(iters: np.ndarray, status: np.ndarray, residuals: np.ndarray) -> dict:
    """Per-shard summary: instances, converged, sum of iterations, max primal / dual residual."""
    res = np.asarray(residuals, dtype=np.float64).reshape(4, -1)
    n = int(np.asarray(iters).size)
    return dict(
        sums=np.array([n, int(np.sum(np.asarray(status) == 1)), int(np.sum(iters))], dtype=np.float64),
        maxs=np.array([float(np.max(res[[0, 2]])) if n else 0.0, float(np.max(res[[1, 3]])) if n else 0.0]),
    )


def allreduce_summary(summary: dict, device=None) -> dict:
    """Combine shard summaries across ranks: ONE all-reduce(sum) + ONE all-reduce(max) of a few
    doubles (latency-only over xGMI). Works without an initialised process group (world size 1)."""
    import torch
    import torch.distributed as dist

    sums = torch.tensor(summary["sums"], dtype=torch.float64, device=device)
    maxs = torch.tensor(summary["maxs"], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        dist.all_reduce(maxs, op=dist.ReduceOp.MAX)
    sums, maxs = sums.cpu().numpy(), maxs.cpu().numpy()
    return dict(instances=int(sums[0]), converged=int(sums[1]), total_iterations=int(sums[2]),
                max_primal_residual=float(maxs[0]), max_dual_residual=float(maxs[1]))
