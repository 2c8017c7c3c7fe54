"""GPU: BASELINE.json's full sizes.  The C oracle is fast enough to check them in full, so the
size-independent properties come on top of an exact comparison."""
import numpy as np
import pytest

from helpers import assert_oracle_parity, native_outputs, oracle_metrics

pytestmark = pytest.mark.gpu


def _run(K, T, G, kind, seed, res=0.5):
    from oracle import oracle as O
    from benchnav_amd import NativeMPPI, synth
    inst = synth.make_instance(G, seed=seed, resolution=res, kind=kind)
    R, state, goal = inst.risk.numpy(), inst.start.numpy(), inst.goal.numpy()
    rng = np.random.default_rng(seed)
    eps = rng.standard_normal((K, T, 2)).astype(np.float32)
    mean = np.clip(rng.standard_normal((T, 2)) * 0.2 + [0.6, 0.0], [0, -1], [1, 1]).astype(np.float32)
    p = O.make_params(K, T, G, res, goal, trig=O.TRIG_SPEC)
    orc = O.solve(p, R, state, mean, eps)
    with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=res, store_controls=True) as pl:
        pl.set_map(R); pl.set_goal(goal); pl.set_mean(mean)
        us, xs = pl.solve(state, eps)
        got = native_outputs(pl, us, xs)
    return got, orc, inst


@pytest.mark.parametrize("K,T,G,kind", [(1024, 50, 256, "smooth"), (1024, 50, 256, "iid"), (8192, 50, 256, "smooth"),
                                         (16384, 100, 512, "smooth"), (16384, 100, 512, "iid")],
                         ids=["c2-smooth", "c2-iid", "c3-size", "c5-smooth", "c5-iid"])
def test_full_size_configs(K, T, G, kind):
    got, orc, inst = _run(K, T, G, kind, seed=11)
    assert_oracle_parity(oracle_metrics(got, orc), ctx=f"K={K} T={T} G={G} {kind}")
    # size-independent properties
    w = got["w"].astype(np.float64)
    assert abs(w.sum() - 1.0) < 1e-4 and (w >= 0).all()
    assert (got["Ustar"][:, 0] >= 0).all() and (got["Ustar"][:, 0] <= 1).all() and (np.abs(got["Ustar"][:, 1]) <= 1).all()
    ext = G * inst.resolution
    last = got["X"][:, -1, :]
    assert (last[:, :2] >= 0).all() and (last[:, :2] <= ext).all() and (np.abs(last[:, 2]) <= np.pi + 1e-6).all()
    step = np.linalg.norm(np.diff(got["X"][:, :, :2], axis=1), axis=2)
    assert step.max() <= 0.1 + 1e-5                      # |dx| <= trav*v*dt <= 0.1 m per step
    # a checksum of checksums over the trajectory batch
    assert np.array_equal(got["X"].view(np.uint32).sum(dtype=np.uint64), orc["X"].view(np.uint32).sum(dtype=np.uint64))
