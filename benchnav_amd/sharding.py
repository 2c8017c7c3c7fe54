"""Multi-GPU: independent planning instances shard across ranks; there is no data-path collective.

The MPPI solve of one instance is a dependent chain (rollout -> softmin -> warm start), so more
GPUs do not shorten one instance's latency; throughput scales by giving every rank (one process
per GPU, torch.distributed, backend "nccl" = RCCL on ROCm) its own map seeds / start-goal
instances (BASELINE.json config 4).  The only exchange is the final gather of per-rank
{solves, seconds}: a few bytes, latency-bound, xGMI bandwidth irrelevant.
"""
from __future__ import annotations

import os
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


def gather_times(local_seconds, device: Optional[torch.device] = None, group=None) -> torch.Tensor:
    """All-gather every rank's vector of timed-region durations (one per repeat) -> (world, repeats) float64 on the CPU.
    The slowest rank bounds each repeat: whole-job time of repeat r = result[:, r].max().  Same transport rules as
    gather_throughput (no group: one row; gloo: CPU tensors; nccl/RCCL: tensors on `device`)."""
    import torch.distributed as dist
    mine = torch.tensor([float(x) for x in local_seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    if not (dist.is_available() and dist.is_initialized()):
        return mine.view(1, -1).cpu()
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, mine, group=group)
    return torch.stack(out).cpu()


# ---------------------------------------------------------------------------------------------
# Optional: ONE solve sharded over the ranks (K=16384 -> 2048 rollouts per GPU).
# The only exchange is an all-gather of the per-workgroup softmin partials (max z, sum e, sum e*u): (2 + 2T) floats
# per 64 rollouts, 207 KB in total at K=16384, T=100 -- one small, latency-bound collective per solve.
# ---------------------------------------------------------------------------------------------
def shard_rollouts(num_samples: int, world_size: int, rank: int):
    """(first rollout, count) of this rank's contiguous shard; shards are multiples of 64 rollouts (one workgroup)."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    if num_samples % 64 != 0:
        raise ValueError("a K-sharded solve needs num_samples to be a multiple of 64")
    groups = shard_instances(num_samples // 64, world_size, rank)
    if not groups:
        raise ValueError(f"rank {rank} would own no rollouts: {num_samples} samples over {world_size} ranks")
    return 64 * groups[0], 64 * len(groups)


def unique_id() -> bytes:
    """128 bytes of ncclGetUniqueId, drawn by the library (bn_dist_unique_id): what rank 0 hands to the ranks of a K-sharded solve."""
    import ctypes as C
    from . import _capi
    buf = (C.c_uint8 * 128)()
    _capi.check(_capi.load().bn_dist_unique_id(C.cast(buf, C.c_void_p)))
    return bytes(buf)


def merge_partials_reference(partials):
    """NumPy statement of the exchange step: merge per-workgroup (max z, sum e, sum e*u[2T]) rows, in row order,
    into (max z, sum e, U* (T,2)).  What bn_mppi_shard_finish_async computes on the device; used by the CPU tests."""
    import numpy as np
    p = np.asarray(partials, np.float64)
    m = p[:, 0].max()
    f = np.exp(p[:, 0] - m)
    S = float((p[:, 1] * f).sum())
    U = (p[:, 2:] * f[:, None]).sum(0) / S
    return float(m), S, U.reshape(-1, 2).astype(np.float32)


class ShardedMPPI:
    """One MPPI solve whose rollouts are split over the ranks of a torch.distributed group (one process per GPU).

    Every rank calls `solve(state)` with the same state and gets the same (U*, X*) -- bit-identical across ranks and
    to the unsharded planner with the same seed -- and keeps the weights / trajectories of its own shard.
    """

    def __init__(self, horizon: int, num_samples: int, grid_size: int, resolution: float, group=None, **planner_kw):
        import torch.distributed as dist
        from .native import NativeMPPI
        self._dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.group = group
        self.world = self._dist.get_world_size(group) if self._dist else 1
        self.rank = self._dist.get_rank(group) if self._dist else 0
        self.first, self.count = shard_rollouts(num_samples, self.world, self.rank)
        # one process per GPU: the planner, its stream and every tensor handed to it live on THIS rank's current device
        self.device = torch.device("cuda", planner_kw.setdefault("device_id", torch.cuda.current_device()))
        planner_kw.setdefault("stream", torch.cuda.current_stream(self.device).cuda_stream)
        self.planner = NativeMPPI(horizon=horizon, num_samples=self.count, grid_size=grid_size, resolution=resolution,
                                  num_instances=1, **planner_kw)
        self.planner.set_rollout_offset(self.first)
        self.T, self.K = horizon, num_samples
        self._counts = [shard_rollouts(num_samples, self.world, r)[1] // 64 for r in range(self.world)]
        self._gathered = torch.empty(sum(self._counts), 2 + 2 * horizon, dtype=torch.float32, device=self.device)
        self._host_backend = bool(self._dist) and self._dist.get_backend(group) != "nccl"
        # Exchange buffers, allocated ONCE (round 3 made a zeros + an empty per solve in the hot loop).  Equal shards on RCCL -- the
        # normal case, K a multiple of 64 x world -- need none: the collective gathers this rank's partial rows straight out of planner
        # memory into `_gathered`, which the tail reads.  Ragged shards are padded to the largest; the gloo rehearsal stages on the host.
        maxc, PS = max(self._counts), 2 + 2 * horizon
        self._direct = (not self._host_backend) and min(self._counts) == maxc
        if self._dist is not None and not self._direct:
            where = "cpu" if self._host_backend else self.device
            self._send = torch.zeros(maxc, PS, dtype=torch.float32, device=where)
            self._recv = torch.empty(self.world * maxc, PS, dtype=torch.float32, device=where)
        self._views = {}                                 # partial rows of the planner's per-solve slots, wrapped once each
        # Equal shards on RCCL: the library enqueues the exchange itself -- rollout kernel, ncclAllGather, tail kernel on the planner's
        # one stream, one C call per solve (bn_mppi_shard_solve_async; round 4 drove three calls and torch's collective stream from
        # here: 60 us per solve on one rank).  The communicator's id travels over the group the job has anyway.
        self._fused = False
        if self._direct and self._dist is not None and os.environ.get("BN_SHARD_TORCH_COLLECTIVE") != "1":
            # ... and every rank must end up on the same path, and none may wait in a collective the others never enter.  The
            # communicator's init (ncclCommInitRank) IS a collective: it blocks until every rank has entered.  So everything a rank can
            # fail on alone -- opening librccl, the handle's state, the exchange buffers, rank 0 drawing the id -- happens first and
            # locally (shard_comm_prepare), the ranks agree on the outcome with one all-reduce over the torch group, and only a
            # unanimous "ready" takes them into the collective; its own outcome is agreed the same way.
            ready, uid = 1, None
            try:
                self.planner.shard_comm_prepare(self.world, self.rank)
                if self.rank == 0:
                    uid = unique_id()
            except Exception:
                ready = 0
            if self._agree(ready, group):
                box = [uid]
                self._dist.broadcast_object_list(box, src=self._dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                ok = 0
                try:
                    self.planner.shard_comm_init(box[0], self.world, self.rank)
                    ok = 1
                except Exception:
                    ok = 0
                self._fused = self._agree(ok, group)

    def _agree(self, flag: int, group) -> bool:
        """True iff every rank of the group passed a non-zero flag (one MIN all-reduce)."""
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=self.device)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MIN, group=group)
        return bool(int(t.item()))

    def _view(self, ptr, shape):
        from .mppi import _DevArray
        return torch.as_tensor(_DevArray(ptr, shape), device=self.device)

    def _partials_tensor(self):
        ptr, n, ps = self.planner.shard_partials()
        v = self._views.get(ptr)
        if v is None:
            v = self._views[ptr] = self._view(ptr, (n, ps))
        return v

    def solve(self, state_dev: torch.Tensor, eps_dev: Optional[torch.Tensor] = None, kind: Optional[int] = None):
        """state_dev: (3,) float32 on the GPU; eps_dev: this rank's slice of the noise or None (Philox in-kernel).
        Three stream-ordered steps, nothing allocated: rollouts of the shard, all-gather of the partial rows, merge + tail."""
        from . import _capi
        if self._fused:
            if eps_dev is None:
                self.planner.shard_solve_async_device(state_dev.data_ptr())
            else:
                self.planner.shard_solve_async_device(state_dev.data_ptr(), eps_dev.data_ptr(), _capi.BN_NOISE_DEVICE_KT2 if kind is None else kind)
            return self
        if eps_dev is None:
            self.planner.shard_rollout_async_device(state_dev.data_ptr())
        else:
            self.planner.shard_rollout_async_device(state_dev.data_ptr(), eps_dev.data_ptr(), _capi.BN_NOISE_DEVICE_KT2 if kind is None else kind)
        mine = self._partials_tensor()
        if self._dist is None:                           # no process group: one rank, nothing to exchange
            self._gathered.copy_(mine)
        elif self._direct:
            self._dist.all_gather_into_tensor(self._gathered, mine, group=self.group)      # RCCL over xGMI with backend "nccl"
        else:
            self._send[:mine.shape[0]].copy_(mine)
            self._dist.all_gather_into_tensor(self._recv, self._send, group=self.group)
            maxc = self._send.shape[0]
            recv = self._recv.view(self.world, maxc, -1)
            row = 0
            for r, c in enumerate(self._counts):         # the valid rows in rank order
                self._gathered[row:row + c].copy_(recv[r, :c])
                row += c
        self.planner.shard_finish_async(self._gathered.data_ptr(), self._gathered.shape[0])
        return self

    def results(self):
        """(U* (T,2), X* (T+1,3)) of the latest solve as device tensors (views of planner memory, stream-ordered)."""
        from . import _capi
        self.planner.flush()                             # (library-enqueued exchange: joins the side-stream tail; enqueue only)
        us = self._view(self.planner.device_buffer(_capi.BN_BUF_USTAR)[0], (self.T, 2))
        xs = self._view(self.planner.device_buffer(_capi.BN_BUF_XSTAR)[0], (self.T + 1, 3))
        return us, xs

    def close(self):
        self.planner.close()
