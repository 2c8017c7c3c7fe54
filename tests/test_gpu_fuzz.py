"""GPU: randomised configurations, HIP vs oracle on the same inputs (bit-exact U, X, cost; tight w, U*, X*).
Every knob of bn_mppi_config moves: K (ragged), T (odd, 1, past a wavefront of columns), G (odd), resolution
(power of two or not), limits (shifted origin, non-square), sigma, lambda, thresholds, action bounds, start
states on / beyond the map edge with headings far outside [-pi, pi], goals anywhere, maps with risks outside
[0, 1]; injected noise in both layouts, with and without the LDS window, with and without materialised
controls, one-launch and two-launch paths."""
import numpy as np
import pytest

from helpers import assert_oracle_parity, native_outputs, oracle_metrics

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    K = int(rng.choice([1, 7, 63, 64, 65, 100, 130, 257, 640, 1000, 2049, 2112, 4100, 4160]))
    T = int(rng.choice([1, 2, 3, 5, 8, 17, 31, 32, 33, 34, 50, 63, 64, 65, 97]))
    if K * T > 160_000:
        T = max(1, 160_000 // K)
    G = int(rng.choice([8, 17, 33, 50, 64, 100, 129, 256]))
    res = float(rng.choice([0.25, 0.5, 1.0, 0.3, 0.1, 0.7, 2.0]))
    x0 = float(rng.choice([0.0, 0.0, -3.5, 10.0, 1.3]))
    y0 = float(rng.choice([0.0, 0.0, 2.25, -7.0]))
    xl, yl = (x0, x0 + G * res), (y0, y0 + G * res)
    R = rng.random((G, G)).astype(np.float32) * rng.choice([0.5, 0.95, 1.3]) - rng.choice([0.0, 0.0, 0.2])
    if rng.random() < 0.3:                                    # smooth-ish field instead of i.i.d.
        c = rng.random((max(2, G // 8), max(2, G // 8))).astype(np.float32)
        R = np.kron(c, np.ones((8, 8), np.float32))[:G, :G]
        R = np.pad(R, ((0, G - R.shape[0]), (0, G - R.shape[1])), mode="edge")
    where = rng.choice(["inside", "edge", "outside"], p=[0.6, 0.25, 0.15])
    if where == "inside":
        pos = np.array([rng.uniform(*xl), rng.uniform(*yl)])
    elif where == "edge":
        pos = np.array([rng.choice([xl[0], xl[1], rng.uniform(*xl)]), rng.choice([yl[0], yl[1]])])
    else:
        pos = np.array([xl[1] + rng.uniform(0, 50), yl[0] - rng.uniform(0, 50)])
    theta = float(rng.choice([rng.uniform(-np.pi, np.pi), rng.uniform(-40, 40), np.pi, -np.pi, 0.0]))
    state = np.array([pos[0], pos[1], theta], np.float32)
    goal = np.array([rng.uniform(xl[0] - 5, xl[1] + 5), rng.uniform(yl[0] - 5, yl[1] + 5)], np.float32)
    sigma = [float(rng.choice([0.5, 0.1, 1.5])), float(rng.choice([0.5, 0.25, 2.0]))]
    lam = float(rng.choice([0.5, 0.05, 3.0, 50.0]))
    thr = float(rng.choice([0.3, 0.0, 0.55, 1.0]))
    u_min = [float(rng.choice([0.0, -0.5])), float(rng.choice([-1.0, -0.3]))]
    u_max = [float(rng.choice([1.0, 0.4])), float(rng.choice([1.0, 2.0]))]
    eps = rng.standard_normal((K, T, 2)).astype(np.float32)
    mean = (rng.standard_normal((T, 2)) * 0.4).astype(np.float32)
    return dict(K=K, T=T, G=G, res=res, xl=xl, yl=yl, R=R, state=state, goal=goal, sigma=sigma, lam=lam, thr=thr, u_min=u_min,
                u_max=u_max, eps=eps, mean=mean, window=bool(rng.random() < 0.8), store_u=bool(rng.random() < 0.7),
                pipeline=bool(rng.random() < 0.7), t2k=bool(rng.random() < 0.5))


@pytest.mark.parametrize("seed", range(128))
def test_random_configuration_matches_oracle(seed):
    _random_configuration(seed, False)


@pytest.mark.parametrize("seed", range(400, 464))
def test_random_configuration_matches_oracle_in_reference_order(seed):
    """The same knobs with BN_FLAG_REFERENCE_ORDER (round 4): the reference-order instantiations of every kernel the knobs select --
    latency / role / ticket, with and without the LDS window, one- and two-launch -- against the oracle's per-step mode; action
    bounds up to |omega| = 2 at dt = 0.1 stay below the 0.5 rad per step where the default arithmetic ends, so the flag decides."""
    _random_configuration(seed, True)


def _random_configuration(seed, ref):
    import torch
    from oracle import oracle as O
    from benchnav_amd import NativeMPPI, _capi
    c = _case(seed)
    K, T, G = c["K"], c["T"], c["G"]
    inv_var = (np.float32(1) / (np.asarray(c["sigma"], np.float32) ** 2)).tolist()
    p = O.make_params(K, T, G, c["res"], c["goal"], thr=c["thr"], lambda_=c["lam"], sigma=c["sigma"], inv_var=inv_var,
                      u_min=c["u_min"], u_max=c["u_max"], x_limits=c["xl"], y_limits=c["yl"], trig=O.TRIG_SPEC_PER_STEP if ref else O.TRIG_SPEC)
    orc = O.solve(p, c["R"], c["state"], c["mean"], c["eps"])
    ctx = {k: v for k, v in c.items() if k not in ("R", "eps", "mean")}
    with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=c["res"], x_limits=list(c["xl"]), y_limits=list(c["yl"]),
                    sigmas=c["sigma"], inv_var=inv_var, lambda_=c["lam"], u_min=c["u_min"], u_max=c["u_max"], stuck_threshold=c["thr"],
                    store_controls=True, lds_window=c["window"], pipeline=c["pipeline"], stream=0, reference_order=ref) as pl:
        assert pl.arithmetic() == ("reference_order" if ref else "spec")
        pl.set_map(c["R"]); pl.set_goal(c["goal"]); pl.set_mean(c["mean"])
        st = torch.from_numpy(c["state"]).cuda()
        if c["t2k"]:
            e = torch.from_numpy(np.ascontiguousarray(c["eps"].transpose(1, 2, 0))).cuda()       # (T,2,K) planner layout
            kind = _capi.BN_NOISE_DEVICE_T2K
        else:
            e = torch.from_numpy(c["eps"]).cuda()
            kind = _capi.BN_NOISE_DEVICE_KT2
        torch.cuda.synchronize()
        pl.solve_async_device(st.data_ptr(), e.data_ptr(), kind)
        pl.sync()
        from benchnav_amd.mppi import _DevArray
        xs = torch.as_tensor(_DevArray(pl.device_buffer(_capi.BN_BUF_XSTAR)[0], (1, T + 1, 3)), device="cuda").cpu().numpy()
        got = native_outputs(pl, pl.get_mean(0)[None], xs)
    m = oracle_metrics(got, orc)
    finite = np.isfinite(orc["w"]).all() and np.isfinite(orc["Ustar"]).all()
    assert m["U_exact"] and m["X_exact"] and m["cost_exact"], (ctx, m)
    if finite:
        # weights: exp() of the device vs libm differs by an ulp or two; tiny lambda amplifies nothing here (z - max)
        assert m["w_abs"] <= 2e-6 or m["w_rel"] <= 1e-4, (ctx, m)
        assert m["Ustar_max"] <= 1e-5 and m["Xstar_max"] <= 5e-5, (ctx, m)


@pytest.mark.parametrize("seed", list(range(200, 232)) + [2327])
def test_random_sampled_slip_configuration_matches_oracle(seed):
    """The same for the sampled-slip mode (injected draws): both kernels (LDS window / global fallback), one- and two-launch.
    (Seed 2327, found by tools/fuzz_sweep.py: a reach whose three windows do not fit the tail's LDS.)"""
    import torch
    from oracle import oracle as O
    from benchnav_amd import NativeMPPI, _capi
    from benchnav_amd.mppi import _DevArray
    c = _case(seed)
    K, T, G = c["K"], c["T"], c["G"]
    rng = np.random.default_rng(seed)
    MU = np.clip(c["R"], -0.2, 1.2).astype(np.float32)
    SG = (rng.random((G, G)) * rng.choice([0.0, 0.05, 0.3])).astype(np.float32)
    zt = rng.standard_normal((K, T)).astype(np.float32); zc = rng.standard_normal((K, T + 1)).astype(np.float32)
    zo = rng.standard_normal(T).astype(np.float32)
    inv_var = (np.float32(1) / (np.asarray(c["sigma"], np.float32) ** 2)).tolist()
    p = O.make_params(K, T, G, c["res"], c["goal"], thr=c["thr"], lambda_=c["lam"], sigma=c["sigma"], inv_var=inv_var,
                      u_min=c["u_min"], u_max=c["u_max"], x_limits=c["xl"], y_limits=c["yl"], trig=O.TRIG_SPEC)
    orc = O.solve_sampled(p, MU, SG, c["state"], c["mean"], c["eps"], zt, zc, zo)
    ctx = {k: v for k, v in c.items() if k not in ("R", "eps", "mean")}
    with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=c["res"], x_limits=list(c["xl"]), y_limits=list(c["yl"]),
                    sigmas=c["sigma"], inv_var=inv_var, lambda_=c["lam"], u_min=c["u_min"], u_max=c["u_max"], stuck_threshold=c["thr"],
                    store_controls=True, lds_window=c["window"], pipeline=c["pipeline"], sampled_slip=True, stream=0) as pl:
        pl.set_map(MU); pl.set_slip_std(SG); pl.set_goal(c["goal"]); pl.set_mean(c["mean"])
        keep = [torch.from_numpy(np.ascontiguousarray(zt.T)).cuda(), torch.from_numpy(np.ascontiguousarray(zc.T)).cuda(), torch.from_numpy(zo).cuda(),
                torch.from_numpy(c["state"]).cuda(), torch.from_numpy(c["eps"]).cuda()]
        torch.cuda.synchronize()
        pl.set_slip_noise(keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr())
        pl.solve_async_device(keep[3].data_ptr(), keep[4].data_ptr(), _capi.BN_NOISE_DEVICE_KT2)
        pl.sync()
        xs = torch.as_tensor(_DevArray(pl.device_buffer(_capi.BN_BUF_XSTAR)[0], (1, T + 1, 3)), device="cuda").cpu().numpy()
        got = native_outputs(pl, pl.get_mean(0)[None], xs)
    m = oracle_metrics(got, orc)
    assert m["U_exact"] and m["X_exact"] and m["cost_exact"], (ctx, m)
    if np.isfinite(orc["w"]).all() and np.isfinite(orc["Ustar"]).all():
        assert m["w_abs"] <= 2e-6 or m["w_rel"] <= 1e-4, (ctx, m)
        assert m["Ustar_max"] <= 1e-5 and m["Xstar_max"] <= 5e-5, (ctx, m)


@pytest.mark.parametrize("seed", range(300, 324))
def test_random_dwa_configuration_matches_oracle(seed):
    """DWA (N3): random geometry, candidate sets (ragged counts, out-of-bounds controls that the transit re-clamps),
    states and sub-goals: trajectories, costs and the argmin bit-exact against the oracle."""
    from oracle import oracle as O
    from benchnav_amd import NativeMPPI
    c = _case(seed)
    G, T = c["G"], min(c["T"], 60)
    rng = np.random.default_rng(seed)
    NA = int(rng.choice([1, 7, 64, 100, 130, 400]))
    acts = np.stack([rng.uniform(-0.3, 1.3, NA), rng.uniform(-1.5, 1.5, NA)], 1).astype(np.float32)
    sub = None if rng.random() < 0.4 else np.array([rng.uniform(*c["xl"]), rng.uniform(*c["yl"])], np.float32)
    p = O.make_params(64, T, G, c["res"], c["goal"], thr=c["thr"], u_min=c["u_min"], u_max=c["u_max"], x_limits=c["xl"], y_limits=c["yl"],
                      trig=O.TRIG_SPEC)
    orc = O.dwa(p, c["R"], c["state"], acts, sub)
    with NativeMPPI(horizon=T, num_samples=64, grid_size=G, resolution=c["res"], x_limits=list(c["xl"]), y_limits=list(c["yl"]),
                    u_min=c["u_min"], u_max=c["u_max"], stuck_threshold=c["thr"], lds_window=c["window"]) as pl:
        pl.set_map(c["R"]); pl.set_goal(c["goal"])
        out = pl.dwa_solve(c["state"], acts, sub)
    assert np.array_equal(out["states"][0], orc["X"]) and np.array_equal(out["costs"][0], orc["cost"]), (seed, NA, T, G)
    assert int(out["best_index"][0]) == orc["best"] and np.array_equal(out["best_states"][0], orc["X"][orc["best"]])
    if np.isfinite(orc["w"]).all():
        assert np.abs(out["weights"][0] - orc["w"]).max() < 2e-6


@pytest.mark.parametrize("seed", range(400, 416))
def test_random_risk_map_matches_oracle(seed):
    """Risk-map precompute (N1): random maps, sample counts (not multiples of 64, past 1024 and 2048) and confidence
    levels incl. 0 and 1, injected draws: VaR bit-exact (selection + lerp), CVaR within summation order."""
    import torch
    from oracle import risk_oracle as RO
    from benchnav_amd.risk import infer_risk_map
    rng = np.random.default_rng(seed)
    G = int(rng.choice([3, 8, 17, 32]))
    n = int(rng.choice([2, 3, 63, 64, 65, 200, 1000, 1024, 1025, 2048, 2500, 4096]))
    q = float(rng.choice([0.0, 1.0, 0.5, 0.9, 0.975, rng.random()]))
    mean = (rng.random((G, G)) * rng.choice([0.5, 1.0, 5.0]) - rng.choice([0.0, 0.3])).astype(np.float32)
    std = (rng.random((G, G)) * rng.choice([0.0, 0.1, 1.0])).astype(np.float32)
    z = rng.standard_normal((n, G, G)).astype(np.float32)
    if rng.random() < 0.3:
        z[: n // 2] = z[n // 2: n // 2 * 2]                       # ties
    for metric in ("var", "cvar"):
        got = infer_risk_map(torch.from_numpy(mean), torch.from_numpy(std), metric, q, num_samples=n, z=torch.from_numpy(z)).cpu().numpy()
        want = RO.infer_risk_map(mean, std, metric, q, z)
        if metric == "var":
            assert np.array_equal(got, want), (seed, G, n, q)
        else:
            both_nan = np.isnan(got) & np.isnan(want)
            assert (both_nan | (np.abs(got - want) <= 2e-6 * np.maximum(1, np.abs(want)))).all(), (seed, G, n, q)
