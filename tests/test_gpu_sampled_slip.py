"""GPU: BASELINE config 3 -- the MPPI solve with the slip sampled per lookup from the latent Normal(mean, std) map
(traversability_model.py:65-69 inside robot_model.py:75 and objectives.py:50).  The reference's own MPPI refuses
this mode, so parity is against the oracle's restatement of the components, bit-exact with the
same injected normals; the Philox stream is checked through properties."""
import numpy as np
import pytest

from helpers import assert_oracle_parity, native_outputs, oracle_metrics

pytestmark = pytest.mark.gpu


def _problem(K, T, G, seed, kind="smooth"):
    from benchnav_amd import synth
    inst = synth.make_instance(G, seed=seed, kind=kind)
    sg = synth.slip_std_map(G, seed=seed).numpy()
    rng = np.random.default_rng(seed)
    eps = rng.standard_normal((K, T, 2)).astype(np.float32)
    mean = np.clip(rng.standard_normal((T, 2)) * 0.2 + [0.6, 0.0], [0, -1], [1, 1]).astype(np.float32)
    z = dict(zt=rng.standard_normal((K, T)).astype(np.float32), zc=rng.standard_normal((K, T + 1)).astype(np.float32),
             zo=rng.standard_normal(T).astype(np.float32))
    return inst, sg, eps, mean, z


def _native(K, T, G, inst, sg, eps, mean, z, **kw):
    import torch
    from benchnav_amd import NativeMPPI
    with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, store_controls=True, sampled_slip=True, **kw) as pl:
        pl.set_map(inst.risk.numpy()); pl.set_slip_std(sg); pl.set_goal(inst.goal.numpy()); pl.set_mean(mean)
        keep = None
        if z is not None:      # planner layout: rollout index fastest
            keep = [torch.from_numpy(np.ascontiguousarray(z["zt"].T)).cuda(), torch.from_numpy(np.ascontiguousarray(z["zc"].T)).cuda(),
                    torch.from_numpy(z["zo"]).cuda()]
            torch.cuda.synchronize()
            pl.set_slip_noise(*(t.data_ptr() for t in keep))
        us, xs = pl.solve(inst.start.numpy(), eps)
        return native_outputs(pl, us, xs)


@pytest.mark.parametrize("K,T,G,kind,window", [(1000, 50, 256, "smooth", True), (192, 7, 64, "iid", True), (8192, 50, 256, "smooth", True),
                                                (64, 1, 64, "iid", True), (1000, 50, 256, "iid", False), (130, 100, 256, "smooth", True)],
                         ids=["ragged", "small-iid", "c3", "T1", "no-window", "T100-fallback"])
def test_sampled_slip_matches_oracle_with_injected_normals(K, T, G, kind, window):
    from oracle import oracle as O
    inst, sg, eps, mean, z = _problem(K, T, G, seed=5, kind=kind)
    p = O.make_params(K, T, G, 0.5, inst.goal.numpy(), trig=O.TRIG_SPEC)
    orc = O.solve_sampled(p, inst.risk.numpy(), sg, inst.start.numpy(), mean, eps, z["zt"], z["zc"], z["zo"])
    got = _native(K, T, G, inst, sg, eps, mean, z, lds_window=window)
    assert_oracle_parity(oracle_metrics(got, orc), ctx=f"sampled K={K} T={T} G={G}")
    # the draws matter: the deterministic solve on the mean map gives other trajectories
    det = O.solve(p, inst.risk.numpy(), inst.start.numpy(), mean, eps)
    assert np.abs(det["X"] - orc["X"]).max() > 1e-3


@pytest.mark.parametrize("K,T,window", [(8192, 50, True), (300, 33, True), (300, 34, False)], ids=["c3", "odd-T", "no-window"])
def test_philox_slip_solve_matches_oracle_on_the_regenerated_draws(K, T, window):
    """The library's own draws (Philox, in-kernel): bn_mppi_get_slip_noise regenerates them, the oracle consumes
    them, and the solve must agree exactly as with injected draws -- control noise from the Philox stream too."""
    from oracle import oracle as O
    from benchnav_amd import NativeMPPI
    G = 256
    inst, sg, _, mean, _ = _problem(K, T, G, seed=21)
    with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, store_controls=True, sampled_slip=True, seed=99,
                    lds_window=window) as pl:
        pl.set_map(inst.risk.numpy()); pl.set_slip_std(sg); pl.set_goal(inst.goal.numpy()); pl.set_mean(mean)
        us, xs = pl.solve(inst.start.numpy())
        got = native_outputs(pl, us, xs)
        n = pl.solve_count()
        eps = pl.philox_noise(n - 1)
        zt, zc, zo = pl.slip_noise(n - 1)
    for z in (zt, zc):
        assert abs(float(z.mean())) < 0.02 and abs(float(z.std()) - 1) < 0.02
    p = O.make_params(K, T, G, 0.5, inst.goal.numpy(), trig=O.TRIG_SPEC)
    orc = O.solve_sampled(p, inst.risk.numpy(), sg, inst.start.numpy(), mean, eps, zt, zc, zo)
    assert_oracle_parity(oracle_metrics(got, orc), ctx=f"philox sampled K={K} T={T}")


def test_zero_std_reduces_to_the_deterministic_planner():
    """z * 0 + mean == mean exactly: with a zero std map the sampled kernel and the five-wave inference kernel
    must agree bit for bit on every trajectory and cost (two independent kernels, one arithmetic)."""
    from benchnav_amd import NativeMPPI
    K, T, G = 8192, 50, 256
    inst, sg, eps, mean, _ = _problem(K, T, G, seed=9)
    got = _native(K, T, G, inst, np.zeros_like(sg), eps, mean, None)
    with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, store_controls=True) as pl:
        pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy()); pl.set_mean(mean)
        us, xs = pl.solve(inst.start.numpy(), eps)
        ref = native_outputs(pl, us, xs)
    for k in ("U", "X", "cost", "Xstar"):
        assert np.array_equal(got[k], ref[k]), k
    assert np.abs(got["w"] - ref["w"]).max() < 1e-6 and np.abs(got["Ustar"] - ref["Ustar"]).max() < 1e-6


def test_philox_slip_stream_is_reproducible_and_seeded():
    K, T, G = 2048, 50, 256
    inst, sg, eps, mean, _ = _problem(K, T, G, seed=3)
    a = _native(K, T, G, inst, sg, eps, mean, None, seed=7)
    b = _native(K, T, G, inst, sg, eps, mean, None, seed=7)
    c = _native(K, T, G, inst, sg, eps, mean, None, seed=8)
    assert np.array_equal(a["X"], b["X"]) and np.array_equal(a["cost"], b["cost"])
    assert np.abs(a["X"] - c["X"]).max() > 1e-3
    assert np.array_equal(a["U"], c["U"])                 # injected control noise: identical perturbed controls
    w = a["w"].astype(np.float64)
    assert abs(w.sum() - 1) < 1e-4 and np.isfinite(a["cost"]).all()
    # per-step displacement = trav*v*dt with trav = 1 - clamp(N(mu, sd)): mean over rollouts sits near the mean-map value
    step = np.linalg.norm(np.diff(a["X"][:, :, :2], axis=1), axis=2)
    assert step.max() <= 0.1 + 1e-5


def test_solve_without_std_map_is_refused():
    from benchnav_amd import NativeMPPI
    with NativeMPPI(horizon=5, num_samples=64, grid_size=32, resolution=0.5, sampled_slip=True) as pl:
        pl.set_map(np.zeros((32, 32), np.float32)); pl.set_goal([8.0, 8.0])
        with pytest.raises(RuntimeError, match="set_slip_std"):
            pl.solve(np.array([4.0, 4.0, 0.0], np.float32))
    with NativeMPPI(horizon=5, num_samples=64, grid_size=32, resolution=0.5) as pl:
        with pytest.raises(RuntimeError, match="SAMPLED_SLIP"):
            pl.set_slip_std(np.zeros((32, 32), np.float32))


def test_sampled_slip_against_the_reference_component_fixture():
    """The `sampled` golden fixture (reference observation-mode components + captured draws) through the C ABI."""
    import torch
    from helpers import TOL_REF, assert_within, load_case, native_planner_for, parity_metrics
    fx = load_case("sampled")
    d = dict(fx); d["x_stride"] = 1
    for k in ("U", "X", "cost", "w", "Ustar", "Xstar"):
        d[f"{k}_0"] = fx[k]
    with native_planner_for(fx, sampled_slip=True) as pl:
        pl.set_map(fx["MU"]); pl.set_slip_std(fx["SG"]); pl.set_goal(fx["goal"]); pl.set_mean(fx["mean"])
        keep = [torch.from_numpy(np.ascontiguousarray(fx["zt"].T)).cuda(), torch.from_numpy(np.ascontiguousarray(fx["zc"].T)).cuda(),
                torch.from_numpy(fx["zo"]).cuda()]
        torch.cuda.synchronize()
        pl.set_slip_noise(*(t.data_ptr() for t in keep))
        us, xs = pl.solve(fx["state"], fx["eps"])
        got = native_outputs(pl, us, xs)
    m = parity_metrics(got, d, 0)
    assert_within(m, TOL_REF, ctx="sampled fixture")
    assert m["X_max"] < 2e-5, m
