"""One-off sweep of the rollout-sharded solve (tests/test_gpu_kshard.py's single-process harness) over random K (multiples of 64),
horizons, world sizes, geometry and both arithmetics: every shard bit-identical to the unsharded solve.
    python tools/fuzz_shard.py 0 200"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from benchnav_amd import NativeMPPI, _capi, synth
from benchnav_amd.sharding import shard_rollouts
from benchnav_amd.mppi import _DevArray


def run(seed):
    rng = np.random.default_rng(70_000 + seed)
    world = int(rng.choice([2, 3, 4, 5, 8]))
    K = 64 * int(rng.integers(world, 200))
    T = int(rng.choice([1, 2, 5, 17, 33, 50, 64, 97, 130]))
    G = int(rng.choice([64, 129, 256]))
    res = float(rng.choice([0.5, 0.25, 0.3]))
    ref_order = bool(rng.random() < 0.3)
    inst = synth.make_instance(G, seed=int(rng.integers(0, 1000)), resolution=res)
    mean = np.clip(rng.standard_normal((T, 2)) * 0.2 + [0.6, 0.0], [0, -1], [1, 1]).astype(np.float32)
    kw = dict(horizon=T, grid_size=G, resolution=res, store_controls=True, seed=int(rng.integers(1, 1 << 20)), reference_order=ref_order,
              lambda_=float(rng.choice([0.5, 0.05, 3.0])))
    try:
        with NativeMPPI(num_samples=K, pipeline=False, **kw) as pl:
            pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy()); pl.set_mean(mean)
            us, xs = pl.solve(inst.start.numpy())
            ref = dict(Ustar=us[0], Xstar=xs[0], w=pl.weights(), cost=pl.costs(), X=pl.states(), U=pl.controls())
    except Exception as e:                                      # noqa: BLE001
        return "skip: " + str(e)[:60]
    st = inst.start.cuda()
    planners, parts = [], []
    try:
        for r in range(world):
            first, count = shard_rollouts(K, world, r)
            pl = NativeMPPI(num_samples=count, stream=0, **kw)
            planners.append((pl, first, count))
            pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy()); pl.set_mean(mean); pl.set_rollout_offset(first)
            pl.shard_rollout_async_device(st.data_ptr())
            ptr, n, ps = pl.shard_partials()
            parts.append(torch.as_tensor(_DevArray(ptr, (n, ps)), device="cuda"))
        gathered = torch.cat(parts).contiguous()
        for pl, first, count in planners:
            pl.shard_finish_async(gathered.data_ptr(), gathered.shape[0])
            pl.sync()
            us = pl.get_mean(0)
            xs = torch.as_tensor(_DevArray(pl.device_buffer(_capi.BN_BUF_XSTAR)[0], (T + 1, 3)), device="cuda").cpu().numpy()
            sl = slice(first, first + count)
            ok = (np.array_equal(pl.controls(), ref["U"][sl]) and np.array_equal(pl.states(), ref["X"][sl]) and np.array_equal(pl.costs(), ref["cost"][sl])
                  and np.array_equal(us, ref["Ustar"]) and np.array_equal(xs, ref["Xstar"]) and np.array_equal(pl.weights(), ref["w"][sl]))
            if not ok:
                return f"MISMATCH world={world} K={K} T={T} G={G} res={res} ref={ref_order} shard at {first}"
    except Exception as e:                                      # noqa: BLE001
        return f"ERROR world={world} K={K} T={T} G={G} res={res} ref={ref_order}: " + repr(e)[:200]
    finally:
        for pl, _, _ in planners: pl.close()
    return "ok"


if __name__ == "__main__":
    tally = {}
    for seed in range(int(sys.argv[1]), int(sys.argv[2])):
        r = run(seed)
        key = r.split(":")[0].split(" world")[0]
        tally[key] = tally.get(key, 0) + 1
        if not (r.startswith("ok") or r.startswith("skip")): print(seed, r, flush=True)
    print("tally:", tally)
