"""Multi-GPU: independent planning instances shard across ranks; there is no data-path collective.

The MPPI solve of one instance is a dependent chain (rollout -> softmin -> warm start), so more
GPUs do not shorten one instance's latency; throughput scales by giving every rank (one process
per GPU, torch.distributed, backend "nccl" = RCCL on ROCm) its own map seeds / start-goal
instances (BASELINE.json config 4).  The only exchange is the final gather of per-rank
{solves, seconds}: a few bytes, latency-bound, xGMI bandwidth irrelevant.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch


def shard_instances(num_instances: int, world_size: int, rank: int) -> List[int]:
    """Contiguous, balanced split of instance ids 0..num_instances-1; the first
    `num_instances % world_size` ranks get one extra instance."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    if num_instances < 0:
        raise ValueError("num_instances must be >= 0")
    base, extra = divmod(num_instances, world_size)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def gather_throughput(local_solves: int, local_seconds: float, device: Optional[torch.device] = None,
                      group=None) -> Dict[str, float]:
    """All-gather every rank's (solves, seconds); whole-job throughput = sum(solves) / max(seconds).

    Works without an initialised process group (single process) and with gloo (CPU tensors) or
    nccl/RCCL (tensors on `device`).
    """
    import torch.distributed as dist
    mine = torch.tensor([float(local_solves), float(local_seconds)], dtype=torch.float64,
                        device=device if device is not None else "cpu")
    if not (dist.is_available() and dist.is_initialized()):
        rows = mine.view(1, 2)
    else:
        world = dist.get_world_size(group)
        out = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(out, mine, group=group)
        rows = torch.stack(out)
    rows = rows.cpu()
    total = float(rows[:, 0].sum())
    tmax = float(rows[:, 1].max())
    return {"total_solves": total, "max_seconds": tmax, "solves_per_s": total / tmax if tmax > 0 else 0.0,
            "per_rank_solves": rows[:, 0].tolist(), "per_rank_seconds": rows[:, 1].tolist()}
