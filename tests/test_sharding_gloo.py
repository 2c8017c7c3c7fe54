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


# ---- K-sharded solve: the exchange step (all-gather of softmin partials, merge in rank order) -----------------
def _kshard_worker(rank, world, port, q):
    import numpy as np
    from benchnav_amd.sharding import merge_partials_reference, shard_rollouts
    from oracle import oracle as O
    from benchnav_amd import synth
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        K, T, G, lam = 320, 12, 64, 0.5
        inst = synth.make_instance(G, seed=2)
        rng = np.random.default_rng(0)
        eps = rng.standard_normal((K, T, 2)).astype(np.float32)
        mean = np.zeros((T, 2), np.float32)
        p = O.make_params(K, T, G, 0.5, inst.goal.numpy(), trig=O.TRIG_SPEC)
        full = O.solve(p, inst.risk.numpy(), inst.start.numpy(), mean, eps)          # the checker: every rank can afford it here
        first, count = shard_rollouts(K, world, rank)
        rows = []
        for g0 in range(first, first + count, 64):                                    # what a rank's workgroups publish
            z = -full["cost"][g0:g0 + 64].astype(np.float64) / lam
            e = np.exp(z - z.max())
            rows.append(np.concatenate([[z.max(), e.sum()], (e[:, None] * full["U"][g0:g0 + 64].reshape(64, -1)).sum(0)]))
        mine = torch.tensor(np.stack(rows), dtype=torch.float64)
        counts = [shard_rollouts(K, world, r)[1] // 64 for r in range(world)]
        send = torch.zeros(max(counts), mine.shape[1], dtype=torch.float64)          # ragged shards (3 + 2 workgroups): pad
        send[:mine.shape[0]] = mine
        recv = torch.empty(world * max(counts), mine.shape[1], dtype=torch.float64)
        dist.all_gather_into_tensor(recv, send)
        recv = recv.view(world, max(counts), -1)
        m, S, U = merge_partials_reference(torch.cat([recv[r, :c] for r, c in enumerate(counts)]).numpy())
        q.put((rank, first, count, float(np.abs(U - full["Ustar"]).max()), U.tobytes()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_world_size_2_kshard_exchange_reproduces_the_full_softmin():
    from benchnav_amd.sharding import shard_rollouts
    assert shard_rollouts(16384, 8, 3) == (3 * 2048, 2048)                            # config 5 over 8 GPUs
    assert [shard_rollouts(320, 2, r) for r in range(2)] == [(0, 192), (192, 128)]
    with pytest.raises(ValueError):
        shard_rollouts(100, 2, 0)
    with pytest.raises(ValueError):
        shard_rollouts(64, 2, 1)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_kshard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=90) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert (got[0][1], got[0][2]) == (0, 192) and (got[1][1], got[1][2]) == (192, 128)
    assert got[0][3] < 2e-6 and got[1][3] < 2e-6           # merged U* == the oracle's full-K U*
    assert got[0][4] == got[1][4]                           # and bit-identical on both ranks
