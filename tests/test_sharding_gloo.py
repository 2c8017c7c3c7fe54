"""CPU: the N > 1 path -- instance sharding and the throughput gather -- with world_size-2 gloo."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from benchnav_amd.sharding import gather_throughput, shard_instances


def test_shards_cover_every_instance_exactly_once():
    for n in (0, 1, 7, 8, 64, 65, 1000):
        for world in (1, 2, 3, 4, 8):
            seen = []
            sizes = []
            for r in range(world):
                s = shard_instances(n, world, r)
                seen += s
                sizes.append(len(s))
            assert seen == list(range(n))
            assert max(sizes) - min(sizes) <= 1
    assert shard_instances(64, 8, 3) == list(range(24, 32))          # config 4: 64 seeds over 8 GPUs
    with pytest.raises(ValueError):
        shard_instances(4, 2, 2)


def test_gather_without_process_group_is_identity():
    g = gather_throughput(100, 2.0)
    assert g["total_solves"] == 100 and g["max_seconds"] == 2.0 and g["solves_per_s"] == 50.0


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = shard_instances(7, world, rank)
        # every "solve" of the fake planner is instantaneous; the rank's time is made up and rank-dependent
        g = gather_throughput(local_solves=10 * len(mine), local_seconds=1.0 + rank)
        q.put((rank, mine, g))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo_gather():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got[0][1] == [0, 1, 2, 3] and got[1][1] == [4, 5, 6]
    for _, _, g in got:                                   # both ranks see the same whole-job numbers
        assert g["total_solves"] == 70.0 and g["max_seconds"] == 2.0 and g["solves_per_s"] == 35.0
        assert g["per_rank_solves"] == [40.0, 30.0]
