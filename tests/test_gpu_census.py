"""GPU: the parity census through the C ABI (DESIGN.md "Arithmetic spec and parity tiers").

The reference ran MPPI.forward at BASELINE sizes on the oracle's portable noise stream (tests/golden/census_*.npz, written by
make_golden.py).  The HIP path replays every stored solve -- same noise, teacher-forced mean and state -- in its two
arithmetics:
  * default (carried heading vector, fused transit): bit-exact with the oracle's spec mode, and against the REFERENCE every
    rollout is classified -- within the 1e-4 trajectory tolerance, or a cell flip (helpers.census_classify asserts the flip) --
    with the share of flips below the stated bound;
  * BN_FLAG_REFERENCE_ORDER (robot_model.py:86-88 as written): bit-exact with the oracle's per-step mode, no flip in these
    fixtures' 0.16 M rollouts, every tier of SURVEY 8a on every solve.
Also here: the slow path that the flag shares with configurations the role kernels cannot take (long horizons, dt |omega| > 0.5,
limits wider than the grid) -- errors until round 3, solves now.
"""
import numpy as np
import pytest

from helpers import (CENSUS_RATE_BOUND, TOL_CENSUS, assert_oracle_parity, assert_within, census_classify, census_eps, census_oracle_params,
                     census_sampled_draws, census_solves, census_tiers, load_case, native_outputs, oracle_metrics)

pytestmark = pytest.mark.gpu


def _planner(fx, **kw):
    from benchnav_amd import NativeMPPI
    return NativeMPPI(horizon=int(fx["T"]), num_samples=int(fx["K"]), grid_size=int(fx["G"]), resolution=float(fx["res"]),
                      sigmas=fx["sigmas"].tolist(), lambda_=float(fx["lam"]), stuck_threshold=float(fx["thr"]), store_controls=True, **kw)


@pytest.mark.parametrize("name", ["census_c2", "census_c5"])
@pytest.mark.parametrize("arith", ["spec", "reference_order"])
def test_census_through_the_c_abi(name, arith):
    from oracle import oracle as O
    fx = load_case(name)
    K, T = int(fx["K"]), int(fx["T"])
    trig = O.TRIG_SPEC if arith == "spec" else O.TRIG_SPEC_PER_STEP
    beyond = total = within4 = outliers = 0
    with _planner(fx, reference_order=(arith == "reference_order")) as pl:
        assert pl.arithmetic() == arith and pl.launches_per_solve() == 1           # both arithmetics run on every kernel (round 4)
        last_map = None
        for mi, i, key in census_solves(fx):
            if mi != last_map:
                pl.set_map(fx[f"R_{mi}"]); pl.set_goal(fx[f"goal_{mi}"]); last_map = mi
            eps = census_eps(fx, mi, i)
            pl.set_mean(fx[f"mean_{key}"])                              # teacher-forced
            us, xs = pl.solve(fx[f"state_{key}"], eps)
            got = native_outputs(pl, us, xs)
            # (1) the kernel against the oracle in the same arithmetic: bit-exact trajectories, controls, costs
            if i == 0 or name == "census_c5":
                orc = O.solve(census_oracle_params(fx, mi, trig), fx[f"R_{mi}"], fx[f"state_{key}"], fx[f"mean_{key}"], eps)
                assert_oracle_parity(oracle_metrics(got, orc), ctx=f"{name} {key} {arith}")
            # (2) against the reference: classify every rollout, flips are asserted to BE flips
            c = census_classify(fx, key, got["X"])
            beyond += len(c["beyond"]); total += K; within4 += int((c["pos_ulp"] <= 4).sum())
            m = census_tiers(got, fx, key, c["beyond"])
            outliers += int(round(m["cost_outlier_frac"] * K))
            assert_within(m, TOL_CENSUS, ctx=f"{name} {key} {arith}")
    assert beyond / total <= CENSUS_RATE_BOUND[arith][T], f"{name} {arith}: {beyond}/{total} rollouts beyond the trajectory tolerance"
    assert within4 / total >= (0.999 if arith == "spec" else 0.9999)
    if arith == "reference_order":
        assert beyond == 0, "the reference-order arithmetic had no flip in these fixtures at capture"
    print(f"{name} {arith}: {beyond}/{total} rollouts beyond 1e-4 (cell flips), {total - within4} beyond 4 ulp, {outliers} cost outliers")


@pytest.mark.parametrize("arith", ["spec", "reference_order"])
def test_census_sampled_slip_configs2_size(arith):
    """BASELINE configs[2] at full size against the REFERENCE's observation-mode components (census_c3), both arithmetics."""
    import torch
    from oracle import oracle as O
    fx = load_case("census_c3")
    K, T = int(fx["K"]), int(fx["T"])
    zt, zc, zo = census_sampled_draws(int(fx["noise_seed_0"]), K, T)
    eps = census_eps(fx, 0, 0)
    with _planner(fx, sampled_slip=True, reference_order=(arith == "reference_order")) as pl:
        pl.set_map(fx["MU"]); pl.set_slip_std(fx["SG"]); pl.set_goal(fx["goal_0"]); pl.set_mean(fx["mean_0_0"])
        keep = [torch.from_numpy(np.ascontiguousarray(zt.T)).cuda(), torch.from_numpy(np.ascontiguousarray(zc.T)).cuda(), torch.from_numpy(zo).cuda()]
        torch.cuda.synchronize()
        pl.set_slip_noise(*(t.data_ptr() for t in keep))
        us, xs = pl.solve(fx["state_0_0"], eps)
        got = native_outputs(pl, us, xs)
    trig = O.TRIG_SPEC if arith == "spec" else O.TRIG_SPEC_PER_STEP
    orc = O.solve_sampled(census_oracle_params(fx, 0, trig), fx["MU"], fx["SG"], fx["state_0_0"], fx["mean_0_0"], eps, zt, zc, zo)
    assert_oracle_parity(oracle_metrics(got, orc), ctx=f"census_c3 {arith}")
    c = census_classify(fx, "0_0", got["X"])
    assert len(c["beyond"]) / K <= CENSUS_RATE_BOUND[arith][50] and (c["pos_ulp"] <= 4).mean() >= 0.999
    assert_within(census_tiers(got, fx, "0_0", c["beyond"]), TOL_CENSUS, ctx=f"census_c3 {arith}")


# ---- the slow path: configurations that were errors until round 3 ------------------------------------------------------------------
def _random_problem(K, T, G, res, seed, x_limits=None):
    from benchnav_amd import synth
    rng = np.random.default_rng(seed)
    R = synth.iid_risk_map(G, seed).numpy()
    ext = G * res
    state = np.array([0.3 * ext, 0.4 * ext, 0.7], np.float32)
    goal = np.array([0.7 * ext, 0.6 * ext], np.float32)
    eps = rng.standard_normal((K, T, 2)).astype(np.float32)
    mean = np.clip(rng.standard_normal((T, 2)) * 0.2 + [0.6, 0.0], [0, -1], [1, 1]).astype(np.float32)
    return R, state, goal, eps, mean


@pytest.mark.parametrize("K,T,G,res,lean", [(256, 400, 64, 0.5, False), (130, 700, 128, 0.25, False), (192, 400, 64, 0.5, True)],
                         ids=["T400", "T700-window-too-big", "T400-lean"])
def test_long_horizons_take_the_slow_path(K, T, G, res, lean):
    """MPPI.__init__ takes any horizon (mppi.py:25).  Beyond the role kernels' LDS the one-wave kernel + tail serve the solve in
    the DEFAULT arithmetic: bit-exact against the oracle's spec mode."""
    from benchnav_amd import NativeMPPI
    from oracle import oracle as O
    R, state, goal, eps, mean = _random_problem(K, T, G, res, seed=T)
    with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=res, store_controls=True, lean=lean) as pl:
        assert pl.arithmetic() == "spec" and pl.launches_per_solve() == 2
        pl.set_map(R); pl.set_goal(goal); pl.set_mean(mean)
        us, xs = pl.solve(state, eps)
        got = native_outputs(pl, us, xs)
        top_s, top_w = pl.top_samples(5)
    orc = O.solve(O.make_params(K, T, G, res, goal, trig=O.TRIG_SPEC), R, state, mean, eps)
    assert_oracle_parity(oracle_metrics(got, orc), ctx=f"slow path T={T}")
    order = np.argsort(-orc["w"], kind="stable")[:5]
    assert np.array_equal(top_s, orc["X"][order]) or np.allclose(top_w, orc["w"][order], atol=1e-6)


@pytest.mark.parametrize("dt,wmax", [(1.0, 1.0), (0.1, 8.0), (2.5, 1.2)], ids=["dt1", "omega8", "dt2.5"])
def test_large_heading_steps_select_the_reference_order(dt, wmax):
    """transit takes any delta_t (robot_model.py:60).  dt * max|omega| > 0.5 is beyond the carried rotation's polynomials: the
    handle takes the reference-order arithmetic by itself -- bit-exact against the oracle's per-step mode, DWA included."""
    from benchnav_amd import NativeMPPI
    from oracle import oracle as O
    K, T, G, res = 320, 30, 64, 0.5
    R, state, goal, eps, mean = _random_problem(K, T, G, res, seed=11)
    kw = dict(u_min=(0.0, -wmax), u_max=(1.0, wmax), dt=dt)
    with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=res, store_controls=True, **kw) as pl:
        assert pl.arithmetic() == "reference_order"
        pl.set_map(R); pl.set_goal(goal); pl.set_mean(mean)
        us, xs = pl.solve(state, eps * 3.0)                            # wide noise: the angular bound is reached
        got = native_outputs(pl, us, xs)
        acts = np.stack(np.meshgrid(np.linspace(0, 1, 6), np.linspace(-wmax, wmax, 7), indexing="ij"), -1).reshape(-1, 2).astype(np.float32)
        d = pl.dwa_solve(state, acts)
    p = O.make_params(K, T, G, res, goal, trig=O.TRIG_SPEC_PER_STEP, **kw)
    orc = O.solve(p, R, state, mean, eps * 3.0)
    assert_oracle_parity(oracle_metrics(got, orc), ctx=f"dt={dt} wmax={wmax}")
    od = O.dwa(p, R, state, acts)
    assert np.array_equal(d["states"][0], od["X"]) and np.array_equal(d["costs"][0], od["cost"]) and int(d["best_index"][0]) == od["best"]


@pytest.mark.parametrize("arith", ["spec", "reference_order"])
def test_limits_wider_than_the_grid_keep_the_index_clamp(arith):
    """x/y_limits spanning MORE cells than the grid has (no reference GridMap does, the C ABI allows it): positions clamped to the
    upper limit have raw cells beyond the window's guard row, so the handle gathers from global memory with the reference's index
    clamp (grid_map.py:209) instead of the clamp-free window.  Bit-exact against the oracle, whose lookup clamps."""
    from benchnav_amd import NativeMPPI
    from oracle import oracle as O
    K, T, G, res = 256, 40, 32, 0.5
    R, _, _, eps, mean = _random_problem(K, T, G, res, seed=3)
    lim = (0.0, 20.0)                                                  # 40 cells of 0.5 m over a 32-cell grid
    state = np.array([19.2, 18.9, 0.6], np.float32)                    # heading into the corner: clamped within a few steps
    goal = np.array([10.0, 10.0], np.float32)
    with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=res, x_limits=lim, y_limits=lim, store_controls=True,
                    reference_order=(arith == "reference_order")) as pl:
        pl.set_map(R); pl.set_goal(goal); pl.set_mean(mean)
        us, xs = pl.solve(state, eps)
        got = native_outputs(pl, us, xs)
    trig = O.TRIG_SPEC if arith == "spec" else O.TRIG_SPEC_PER_STEP
    orc = O.solve(O.make_params(K, T, G, res, goal, x_limits=lim, y_limits=lim, trig=trig), R, state, mean, eps)
    assert (orc["X"][:, :, :2].max() > 19.99), "the case must reach the upper limits"
    assert_oracle_parity(oracle_metrics(got, orc), ctx=f"wide limits {arith}")


def test_reference_order_serves_every_noise_source_and_the_lean_reroll():
    """The flag through the remaining entry points: Philox noise (regenerated for the oracle), device-resident noise in both
    layouts, a warm-started chain enqueued with solve_n, lean mode's re-rolled rows."""
    import torch
    from benchnav_amd import NativeMPPI, _capi
    from oracle import oracle as O
    K, T, G, res = 1000, 50, 128, 0.5
    R, state, goal, eps, mean = _random_problem(K, T, G, res, seed=8)
    p = O.make_params(K, T, G, res, goal, trig=O.TRIG_SPEC_PER_STEP)
    sd = torch.from_numpy(state).cuda()
    for lean in (False, True):
        with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=res, reference_order=True, lean=lean, seed=77) as pl:
            pl.set_map(R); pl.set_goal(goal); pl.set_mean(mean)
            n = 3
            pl.solve_n_async_device(n, sd.data_ptr())                  # Philox, three warm-started solves
            pl.sync()
            X, c, w = pl.states(), pl.costs(), pl.weights()
            m = mean
            for i in range(n):
                orc = O.solve(p, R, state, m, pl.philox_noise(i))
                m_next = orc["Ustar"]
                if i < n - 1:
                    m = m_next
            # the chain's means agree to ~1e-6 (fp32 merge order vs the oracle's fp64), so the last solve is compared loosely ...
            assert np.abs(X - orc["X"]).max() < 1e-3
            # ... and exactly when the oracle starts from the mean the kernel really used
            pl.set_mean(mean)
            us, xs = pl.solve(state, eps)
            orc1 = O.solve(p, R, state, mean, eps)
            got = dict(U=orc1["U"], X=pl.states(), cost=pl.costs(), w=pl.weights(), Ustar=us[0], Xstar=xs[0])
            assert_oracle_parity(oracle_metrics(got, orc1), ctx=f"reference order lean={lean}")
            for layout, kind in ((eps, _capi.BN_NOISE_DEVICE_KT2), (np.ascontiguousarray(eps.transpose(1, 2, 0)), _capi.BN_NOISE_DEVICE_T2K)):
                ed = torch.from_numpy(layout).cuda()
                torch.cuda.synchronize()
                pl.set_mean(mean)
                pl.solve_async_device(sd.data_ptr(), ed.data_ptr(), kind)
                pl.sync()
                assert np.array_equal(pl.states(), orc1["X"]) and np.array_equal(pl.costs(), orc1["cost"])


def test_drop_in_class_in_reference_order():
    """benchnav_amd.MPPI(reference_order=True) on the c1 fixture (BASELINE configs[0], the reference's own noise): the arithmetic
    the class reports, and every rollout within the UN-tiered trajectory tolerance of the reference, no cost outlier."""
    import torch
    from helpers import TOL_REF, mppi_for_fixture, parity_metrics
    fx = load_case("c1_basic")
    solver = mppi_for_fixture(fx, noise="torch", reference_order=True)
    assert solver.arithmetic == "reference_order"
    assert mppi_for_fixture(fx, noise="torch").arithmetic == "spec"
    for i in range(int(fx["n_solves"])):
        solver._previous_action_seq = torch.from_numpy(fx[f"mean_{i}"])
        with torch.no_grad():
            U, X = solver.solve_with_noise(torch.tensor(fx[f"state_{i}"]), torch.from_numpy(fx[f"eps_{i}"]))
        got = dict(U=solver._perturbed_action_seqs.cpu().numpy(), X=solver._state_seq_batch.cpu().numpy(), cost=solver._costs.cpu().numpy(),
                   w=solver._weights.cpu().numpy(), Ustar=U.cpu().numpy(), Xstar=X[0].cpu().numpy())
        m = parity_metrics(got, fx, i)
        assert_within(m, TOL_REF, ctx=f"c1_basic solve {i} reference order")
        assert m["cost_outliers"] == 0 and m["X_max"] <= 1e-5


@pytest.mark.parametrize("kernel,B,K,T", [("role", 3, 1000, 50), ("wave", 3, 1000, 50), ("lat", 2, 512, 33), ("role", 1, 6000, 20), ("auto", 2, 128, 1)],
                         ids=["role-B3", "wave-B3", "lat-B2", "ticket-K6000", "T1"])
def test_reference_order_on_every_kernel_family(kernel, B, K, T):
    """Every kernel family has its own instantiation in the reference's operation order (rollout_role_ref_*.hip, rollout_wave_ref.hip):
    forced one at a time, B instances, warm-started second solve included -- bit-exact against the oracle's per-step mode."""
    import torch
    from benchnav_amd import NativeMPPI, _capi, synth
    from oracle import oracle as O
    G, res = 128, 0.5
    insts = [synth.make_instance(G, seed=50 + b, jitter=True) for b in range(B)]
    rng = np.random.default_rng(K + T)
    eps = rng.standard_normal((2, B, K, T, 2)).astype(np.float32)
    states = np.stack([it.start.numpy() for it in insts]).astype(np.float32)
    with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=res, num_instances=B, store_controls=True, kernel=kernel,
                    reference_order=True) as pl:
        assert pl.arithmetic() == "reference_order" and pl.launches_per_solve() == 1
        for b, it in enumerate(insts):
            pl.set_map(it.risk.numpy(), b); pl.set_goal(it.goal.numpy(), b)
        us0, _ = pl.solve(states, eps[0])
        us1, xs1 = pl.solve(states, eps[1])                             # warm-started from the kernel's own U*
        for b, it in enumerate(insts):
            p = O.make_params(K, T, G, res, it.goal.numpy(), trig=O.TRIG_SPEC_PER_STEP)
            orc = O.solve(p, it.risk.numpy(), states[b], us0[b], eps[1, b])
            assert_oracle_parity(oracle_metrics(native_outputs(pl, us1, xs1, b), orc), ctx=f"{kernel} instance {b}")
