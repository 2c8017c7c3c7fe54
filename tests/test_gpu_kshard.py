"""GPU: one solve sharded over ranks by rollouts (optional row).  A single process plays every
rank (one handle per shard on the same device, torch.cat as the all-gather); the exchange over torch.distributed
is covered by tests/test_sharding_gloo.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _unsharded(K, T, G, inst, mean, seed, eps=None, **kw):
    import torch
    from benchnav_amd import NativeMPPI
    with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, store_controls=True, seed=seed, pipeline=False, **kw) as pl:
        pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy()); pl.set_mean(mean)
        us, xs = pl.solve(inst.start.numpy(), eps)
        return dict(Ustar=us[0], Xstar=xs[0], w=pl.weights(), cost=pl.costs(), X=pl.states(), U=pl.controls())


@pytest.mark.parametrize("K,T,world,ref_order", [(4096, 50, 2, False), (4096, 50, 4, False), (16384, 100, 8, False), (1024, 33, 3, False), (4096, 50, 4, True)],
                         ids=["w2", "w4", "c5-w8", "ragged-w3", "w4-reference-order"])
def test_sharded_solve_is_bit_identical_to_the_unsharded_one(K, T, world, ref_order):
    import torch
    from benchnav_amd import NativeMPPI, _capi, synth
    from benchnav_amd.sharding import shard_rollouts
    G = 512 if K > 8192 else 256
    inst = synth.make_instance(G, seed=4)
    rng = np.random.default_rng(4)
    mean = np.clip(rng.standard_normal((T, 2)) * 0.2 + [0.6, 0.0], [0, -1], [1, 1]).astype(np.float32)
    ref = _unsharded(K, T, G, inst, mean, seed=123, reference_order=ref_order)
    st = inst.start.cuda()
    planners, parts = [], []
    for r in range(world):
        first, count = shard_rollouts(K, world, r)
        pl = NativeMPPI(horizon=T, num_samples=count, grid_size=G, resolution=0.5, store_controls=True, seed=123, stream=0, reference_order=ref_order)
        pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy()); pl.set_mean(mean); pl.set_rollout_offset(first)
        pl.shard_rollout_async_device(st.data_ptr())
        ptr, n, ps = pl.shard_partials()
        assert n == count // 64 and ps == 2 + 2 * T
        from benchnav_amd.mppi import _DevArray
        parts.append(torch.as_tensor(_DevArray(ptr, (n, ps)), device="cuda"))
        planners.append((pl, first, count))
    gathered = torch.cat(parts).contiguous()
    assert gathered.shape[0] == K // 64
    outs = []
    for pl, first, count in planners:
        pl.shard_finish_async(gathered.data_ptr(), gathered.shape[0])
        pl.sync()
        us = pl.get_mean(0)
        xs = torch.as_tensor(_DevArray(pl.device_buffer(_capi.BN_BUF_XSTAR)[0], (T + 1, 3)), device="cuda").cpu().numpy()
        outs.append((us, xs, pl.weights(), pl.costs(), pl.states(), pl.controls()))
    for (pl, first, count), (us, xs, w, c, X, U) in zip(planners, outs):
        sl = slice(first, first + count)
        assert np.array_equal(U, ref["U"][sl]) and np.array_equal(X, ref["X"][sl]) and np.array_equal(c, ref["cost"][sl]), first
        assert np.array_equal(us, ref["Ustar"]) and np.array_equal(xs, ref["Xstar"]), first      # same merge, same order
        assert np.array_equal(w, ref["w"][sl]), first
    assert abs(sum(float(o[2].astype(np.float64).sum()) for o in outs) - 1.0) < 1e-4
    # second solve: every shard warm-starts from the same U*
    for pl, _, _ in planners:
        assert np.array_equal(pl.get_mean(0), ref["Ustar"])
        pl.close()


def test_shard_protocol_errors():
    from benchnav_amd import NativeMPPI, synth
    import torch
    inst = synth.make_instance(64, seed=1)
    st = inst.start.cuda()
    with NativeMPPI(horizon=10, num_samples=128, grid_size=64, resolution=0.5, stream=0) as pl:
        pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
        with pytest.raises(RuntimeError, match="no sharded solve"):
            pl.shard_partials()
        pl.shard_rollout_async_device(st.data_ptr())
        with pytest.raises(RuntimeError, match="shard_finish"):
            pl.solve_async_device(st.data_ptr())
        ptr, n, ps = pl.shard_partials()
        with pytest.raises(RuntimeError, match="every shard"):
            pl.shard_finish_async(ptr, n - 1)
        pl.shard_finish_async(ptr, n)            # world of one: its own partials
        pl.sync()
        w = pl.weights().astype(np.float64)
        assert abs(w.sum() - 1) < 1e-5
    with NativeMPPI(horizon=10, num_samples=128, grid_size=64, resolution=0.5, num_instances=2, stream=0) as pl:
        pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
        with pytest.raises(RuntimeError, match="one instance"):
            pl.shard_rollout_async_device(torch.stack([st, st]).data_ptr())


def test_library_enqueued_exchange_protocol_and_results():
    """bn_mppi_shard_solve_async (round 5): needs the communicator first, refuses a second one and a rank outside the world; with a
    communicator of one rank (no torch process group involved: the id is drawn and consumed in this process) three warm-started
    solves are bit-identical to the three-call protocol on a second handle, and every result getter sees the side-stream tail."""
    import torch
    from benchnav_amd import NativeMPPI, synth
    from benchnav_amd.sharding import unique_id
    K, T, G = 2048, 33, 128
    inst = synth.make_instance(G, seed=6)
    st = inst.start.cuda()
    with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, seed=5, stream=0) as a, \
         NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, seed=5, stream=0) as b:
        for pl in (a, b):
            pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
        with pytest.raises(RuntimeError, match="shard_comm_init must precede"):
            a.shard_solve_async_device(st.data_ptr())
        uid = unique_id()
        assert len(uid) == 128
        with pytest.raises(RuntimeError, match="outside a world"):
            a.shard_comm_init(uid, 1, 1)
        a.shard_comm_init(uid, 1, 0)
        with pytest.raises(RuntimeError, match="communicator already"):
            a.shard_comm_init(unique_id(), 1, 0)
        for i in range(3):
            a.shard_solve_async_device(st.data_ptr())
            b.shard_rollout_async_device(st.data_ptr())
            ptr, n, ps = b.shard_partials()
            b.shard_finish_async(ptr, n)
            for what in ("get_mean", "weights", "costs"):            # (each getter orders the handle's stream behind the side-stream tail)
                assert np.array_equal(getattr(a, what)(), getattr(b, what)()), (i, what)
        a.sync(); b.sync()
        assert np.array_equal(a.states(), b.states())


def test_library_enqueued_exchange_at_config5_size_takes_the_two_level_merge():
    """BASELINE configs[4] (K=16384, T=100, 512x512): 256 partial rows, so the merge kernel of the library-enqueued exchange is the
    multi-workgroup one -- 16 groups of 16 rows, the last workgroup to take its ticket merges the group rows (shard_merge_kernel, nblk > 64).
    A chain of warm-started solves on a communicator of one rank against (a) the three-call protocol and (b) the UNSHARDED planner:
    U*, X*, weights, costs, mean bit for bit at every step (this path had a rate check only)."""
    import torch
    from benchnav_amd import NativeMPPI, _capi, synth
    from benchnav_amd.sharding import unique_id
    from benchnav_amd.mppi import _DevArray
    K, T, G = 16384, 100, 512
    inst = synth.make_instance(G, seed=2)
    st = inst.start.cuda()

    def outs(pl):
        pl.flush()
        blk = torch.as_tensor(_DevArray(pl.device_buffer(_capi.BN_BUF_USTAR_XSTAR)[0], (T * 2 + (T + 1) * 3,)), device="cuda")
        torch.cuda.synchronize(); pl.sync()
        return blk.cpu().numpy().copy(), pl.weights(), pl.costs(), pl.get_mean()

    with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, seed=5, stream=0) as a, \
         NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, seed=5, stream=0) as b, \
         NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, seed=5, stream=0) as c:
        for pl in (a, b, c):
            pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
        a.shard_comm_prepare(1, 0)
        a.shard_comm_init(unique_id(), 1, 0)
        for i in range(6):                                   # past kSlots = 4: every rotating per-solve slot is reused
            a.shard_solve_async_device(st.data_ptr())
            b.shard_rollout_async_device(st.data_ptr())
            ptr, n, ps = b.shard_partials()
            assert n == 256
            b.shard_finish_async(ptr, n)
            c.forward_async_device(st.data_ptr())
            oa, ob, oc = outs(a), outs(b), outs(c)
            for j in range(4):
                assert np.array_equal(oa[j], ob[j]), (i, j, "fused vs three calls")
                assert np.array_equal(oa[j], oc[j]), (i, j, "fused vs unsharded")
        assert np.array_equal(a.states(), c.states())


def _dist_worker(rank, world, port, q):
    import os
    import torch
    import torch.distributed as dist
    from benchnav_amd import synth
    from benchnav_amd.sharding import ShardedMPPI
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)       # one GPU on the box: RCCL refuses two ranks per device
    try:
        torch.cuda.set_device(0)
        K, T, G = 2112, 50, 256      # 33 workgroups: ragged over two ranks (17 + 16)
        inst = synth.make_instance(G, seed=4)
        sp = ShardedMPPI(T, K, G, 0.5, seed=77)
        sp.planner.set_map(inst.risk.numpy()); sp.planner.set_goal(inst.goal.numpy())
        st = inst.start.cuda()
        for _ in range(3):                                             # warm-started chain of sharded solves
            sp.solve(st)
        us, xs = sp.results()
        torch.cuda.synchronize()
        q.put((rank, sp.first, sp.count, us.cpu().numpy().tobytes(), xs.cpu().numpy().tobytes(), float(sp.planner.weights().astype("float64").sum())))
        dist.barrier()
        sp.close()
    finally:
        dist.destroy_process_group()


def test_two_process_sharded_solve_over_torch_distributed():
    import socket
    import torch
    import torch.multiprocessing as mp
    from benchnav_amd import NativeMPPI, synth
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dist_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    K, T, G = 2112, 50, 256      # 33 workgroups: ragged over two ranks (17 + 16)
    inst = synth.make_instance(G, seed=4)
    with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, seed=77, pipeline=False) as pl:
        pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
        for _ in range(3):
            us, xs = pl.solve(inst.start.numpy())
    assert got[0][3] == got[1][3] == us[0].tobytes()                   # three warm-started solves later: still bit-identical
    assert got[0][4] == got[1][4] == xs[0].tobytes()
    assert abs(got[0][5] + got[1][5] - 1.0) < 1e-4                     # the shards' weights sum to one


_RCCL_ONE_RANK = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from benchnav_amd import NativeMPPI, synth
from benchnav_amd.sharding import ShardedMPPI
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0), rank=0, world_size=1)
K, T, G = 4096, 50, 256
inst = synth.make_instance(G, seed=4)
st = inst.start.cuda()
sh = ShardedMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, seed=77)
assert sh._dist is not None and not sh._host_backend and sh.world == 1
assert sh._fused == (os.environ.get("BN_SHARD_TORCH_COLLECTIVE") != "1"), sh._fused
sh.planner.set_map(inst.risk.numpy()); sh.planner.set_goal(inst.goal.numpy())
outs = []
for i in range(3):                                   # warm-started chain: every solve's exchange goes through all_gather_into_tensor
    sh.solve(st)
    us, xs = sh.results()
    torch.cuda.synchronize(); sh.planner.sync()
    outs.append((us.cpu().numpy().copy(), xs.cpu().numpy().copy()))
sh.close()
with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, seed=77, pipeline=False) as pl:
    pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
    for i in range(3):
        us, xs = pl.solve(inst.start.numpy())
        assert np.array_equal(us[0], outs[i][0]) and np.array_equal(xs[0], outs[i][1]), i
dist.barrier(); dist.destroy_process_group()
print("RCCL-ONE-RANK-OK", dist.is_nccl_available())
"""


@pytest.mark.parametrize("exchange", ["library", "torch"])
def test_sharded_solve_over_an_rccl_group_of_one(exchange):
    """ShardedMPPI with torch.distributed initialised on backend "nccl" (= RCCL) and a world of one -- the branch an 8-GPU job
    takes.  `library`: the exchange is enqueued by the library itself (bn_mppi_shard_solve_async: rollout kernel, ncclAllGather on
    the planner's stream through a communicator of its own, tail kernel; the id travels over the torch group); `torch`: round 4's
    three calls with all_gather_into_tensor (BN_SHARD_TORCH_COLLECTIVE=1, what ragged shards still take).  Either way the chain of
    three warm-started solves is bit-identical to the unsharded planner's.  In a subprocess: the process group must not leak."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "BN_SHARD_TORCH_COLLECTIVE")}
    if exchange == "torch":
        env["BN_SHARD_TORCH_COLLECTIVE"] = "1"
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29611" if exchange == "torch" else "29613", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK, root], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL-ONE-RANK-OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
