"""GPU: the HIP path, called through the C ABI, against the oracle and the reference fixtures.

Tiers (DESIGN.md "Parity tiers"):
  * vs oracle in spec-trig mode on the same inputs: clamped controls U, all K x (T+1) x 3 slots of
    the trajectory batch X and the per-rollout costs are BIT-EXACT; weights / U* / X* within
    TOL_ORACLE (expf and reduction order are implementation-defined);
  * vs the golden fixtures captured from the reference: TOL_REF (fp32 tolerance stated in helpers.py).
"""
import numpy as np
import pytest

from helpers import (CASES, ILL_CONDITIONED, TOL_REF, assert_oracle_parity, assert_within, load_case,
                     native_outputs, native_planner_for, oracle_metrics, oracle_params_for, parity_metrics, tolerance_for)

pytestmark = pytest.mark.gpu


def _oracle():
    from oracle import oracle as O
    return O


@pytest.mark.parametrize("lds_window", [True, False], ids=["lds", "global"])
@pytest.mark.parametrize("name", CASES)
def test_fixture_cases_teacher_forced(name, lds_window):
    O = _oracle()
    fx = load_case(name)
    p = oracle_params_for(fx, O.TRIG_SPEC)
    with native_planner_for(fx, lds_window=lds_window) as pl:
        pl.set_map(fx["R"]); pl.set_goal(fx["goal"])
        for i in range(int(fx["n_solves"])):
            pl.set_mean(fx[f"mean_{i}"])                       # teacher forcing (SURVEY.md 8a vi)
            us, xs = pl.solve(fx[f"state_{i}"], fx[f"eps_{i}"])
            got = native_outputs(pl, us, xs)
            orc = O.solve(p, fx["R"], fx[f"state_{i}"], fx[f"mean_{i}"], fx[f"eps_{i}"])
            assert_oracle_parity(oracle_metrics(got, orc), ctx=f"{name} solve {i}")
            assert_within(parity_metrics(got, fx, i), tolerance_for(name, fx, i), ctx=f"{name} solve {i}")
            assert np.array_equal(pl.get_mean(), us[0]), "warm start must be U* unshifted (mppi.py:217)"


@pytest.mark.parametrize("name", ["c1_basic", "ref5000"])
def test_free_running_warm_start_tracks_the_reference(name):
    """No teacher forcing: the planner's own U* feeds the next mean, as in the reference loop.  ref5000 is the operating point the
    reference states (test/test_mppi.py:121-169: K=5000 -- ragged, ticket merge --, T=50, 64x64 CVaR-0.9 map, three solves)."""
    fx = load_case(name)
    with native_planner_for(fx) as pl:
        pl.set_map(fx["R"]); pl.set_goal(fx["goal"])
        for i in range(int(fx["n_solves"])):
            us, xs = pl.solve(fx[f"state_{i}"], fx[f"eps_{i}"])
            assert np.abs(us[0] - fx[f"Ustar_{i}"]).max() < 1e-3
            assert np.abs(xs[0] - fx[f"Xstar_{i}"]).max() < 1e-3
        assert pl.solve_count() == int(fx["n_solves"])


@pytest.mark.parametrize("B", [1, 3])
def test_pipelined_async_chain_equals_synchronous_chain(B):
    """K <= 2048: consecutive async solves are software-pipelined (one launch each: the launch of solve i
    merges solve i-1's statistics and writes solve i-1's tail).  The results must be bit-identical to
    the same chain run with a flush after every solve and to the two-launch (non-pipelined) path."""
    import torch
    from benchnav_amd import NativeMPPI, _capi
    fx = load_case("c1_basic")
    K, T, G = int(fx["K"]), int(fx["T"]), int(fx["G"])
    n = 6
    rng = np.random.default_rng(3)
    eps = [torch.from_numpy(rng.standard_normal((B, K, T, 2)).astype(np.float32)).cuda() for _ in range(n)]
    states = [torch.from_numpy(np.tile(fx["state_0"], (B, 1)) + 0.3 * i + 0.1 * np.arange(B)[:, None]).float().cuda()
              for i in range(n)]
    torch.cuda.synchronize()
    results = {}
    for mode in ("pipelined", "flushed", "two_launch"):
        with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=float(fx["res"]), num_instances=B,
                        shared_map=True, store_controls=True, pipeline=(mode != "two_launch")) as pl:
            pl.set_map(fx["R"]); pl.set_goal(fx["goal"])
            for i in range(n):
                pl.solve_async_device(states[i].data_ptr(), eps[i].data_ptr(), _capi.BN_NOISE_DEVICE_KT2)
                if mode == "flushed":
                    pl.sync()
            pl.sync()
            results[mode] = [(pl.states(b), pl.costs(b), pl.weights(b), pl.get_mean(b), pl.controls(b)) for b in range(B)]
            assert pl.solve_count() == n
    for mode in ("flushed", "two_launch"):
        for b in range(B):
            for got, ref in zip(results["pipelined"][b], results[mode][b]):
                assert np.array_equal(got, ref), mode
    assert not np.array_equal(results["pipelined"][0][3], np.zeros((T, 2), np.float32))


def test_noise_layouts_and_state_memory_kinds_agree_bitwise():
    import torch
    from benchnav_amd import _capi
    fx = load_case("c1_stuck")
    eps = torch.from_numpy(fx["eps_0"])
    with native_planner_for(fx) as pl:
        pl.set_map(fx["R"]); pl.set_goal(fx["goal"])
        pl.set_mean(fx["mean_0"])
        us0, xs0 = pl.solve(fx["state_0"], fx["eps_0"])
        ref = native_outputs(pl, us0, xs0)
        st = torch.tensor(fx["state_0"], device="cuda")
        for kind, dev_eps in ((_capi.BN_NOISE_DEVICE_KT2, eps.cuda()),
                              (_capi.BN_NOISE_DEVICE_T2K, eps.permute(1, 2, 0).contiguous().cuda())):
            torch.cuda.synchronize()
            pl.set_mean(fx["mean_0"])
            pl.solve_async_device(st.data_ptr(), dev_eps.data_ptr(), kind)
            pl.sync()
            assert np.array_equal(pl.states(), ref["X"]) and np.array_equal(pl.costs(), ref["cost"])
            assert np.array_equal(pl.weights(), ref["w"]) and np.array_equal(pl.get_mean(), ref["Ustar"])


@pytest.mark.parametrize("K,T,G,res", [(1, 1, 8, 1.0), (63, 5, 16, 0.5), (65, 3, 16, 0.25), (130, 64, 40, 0.7), (64, 129, 96, 0.5)])
def test_ragged_and_tiny_sizes_against_oracle(K, T, G, res):
    O = _oracle()
    from benchnav_amd import NativeMPPI
    rng = np.random.default_rng(K * 1000 + T)
    R = (rng.random((G, G)) * 0.95).astype(np.float32)
    ext = G * res
    state = np.array([0.3 * ext, 0.6 * ext, 1.0], np.float32)
    goal = np.array([0.8 * ext, 0.2 * ext], np.float32)
    eps = rng.standard_normal((K, T, 2)).astype(np.float32)
    mean = (rng.random((T, 2)).astype(np.float32) - 0.3)
    p = O.make_params(K, T, G, res, goal, trig=O.TRIG_SPEC)
    orc = O.solve(p, R, state, mean, eps)
    with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=res, store_controls=True) as pl:
        pl.set_map(R); pl.set_goal(goal); pl.set_mean(mean)
        us, xs = pl.solve(state, eps)
        assert_oracle_parity(oracle_metrics(native_outputs(pl, us, xs), orc), ctx=f"K={K} T={T}")


@pytest.mark.parametrize("res,x_lim,y_lim", [(0.5, (-4.0, 12.0), (3.0, 19.0)), (0.3, (-2.4, 7.2), (1.5, 11.1))],
                         ids=["pow2-res-shifted-origin", "general-res-shifted-origin"])
def test_shifted_origin_geometry_against_oracle(res, x_lim, y_lim):
    """x/y_limits whose lower bound is not 0: the index origin and the lower clamp move with it
    (grid_map.py:199-201, robot_model.py:93-94); exercises the kGeoPow2 and kGeoGeneral kernels."""
    O = _oracle()
    from benchnav_amd import NativeMPPI
    K, T, G = 200, 24, 32
    rng = np.random.default_rng(11)
    R = (rng.random((G, G)) * 0.9).astype(np.float32)
    state = np.array([x_lim[0] + 0.2, y_lim[1] - 0.3, 2.2], np.float32)        # near a corner: clamps on both axes
    goal = np.array([x_lim[0] + 9.0, y_lim[0] + 5.0], np.float32)
    eps = rng.standard_normal((K, T, 2)).astype(np.float32)
    p = O.make_params(K, T, G, res, goal, x_limits=x_lim, y_limits=y_lim, trig=O.TRIG_SPEC)
    orc = O.solve(p, R, state, np.zeros((T, 2), np.float32), eps)
    for lds in (True, False):
        with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=res, x_limits=x_lim, y_limits=y_lim,
                        store_controls=True, lds_window=lds) as pl:
            pl.set_map(R); pl.set_goal(goal)
            us, xs = pl.solve(state, eps)
            assert_oracle_parity(oracle_metrics(native_outputs(pl, us, xs), orc), ctx=f"res={res} lds={lds}")


@pytest.mark.parametrize("kernel", ["lat", "role", "wave"])
def test_rollouts_pinned_to_the_upper_map_limits(kernel):
    """A position ON the upper limit has the raw cell G; the reference clamps the index to G-1 (grid_map.py:209).  The kernels'
    reachable window carries a guard row / column for that cell instead of clamping in the chain: start next to
    the upper-right corner, heading out, so that most rollouts spend most steps on x_hi and / or y_hi."""
    O = _oracle()
    from benchnav_amd import NativeMPPI
    K, T, G, res = 256, 30, 64, 0.5
    rng = np.random.default_rng(23)
    R = (rng.random((G, G)) * 0.8).astype(np.float32)
    hi = G * res
    state = np.array([hi - 0.3, hi - 0.2, 0.7], np.float32)
    goal = np.array([hi + 3.0, hi + 3.0], np.float32)
    mean = np.tile(np.array([0.9, 0.0], np.float32), (T, 1))
    eps = rng.standard_normal((K, T, 2)).astype(np.float32)
    p = O.make_params(K, T, G, res, goal, trig=O.TRIG_SPEC)
    orc = O.solve(p, R, state, mean, eps)
    X = np.asarray(orc["X"])
    assert (X[..., 0] >= hi).mean() > 0.2 and (X[..., 1] >= hi).mean() > 0.2, "the case does not reach the limits"
    with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=res, store_controls=True, kernel=kernel) as pl:
        pl.set_map(R); pl.set_goal(goal); pl.set_mean(mean)
        us, xs = pl.solve(state, eps)
        assert_oracle_parity(oracle_metrics(native_outputs(pl, us, xs), orc), ctx=f"kernel={kernel}")


def test_solve_n_async_equals_a_python_loop():
    import torch
    from benchnav_amd import _capi
    fx = load_case("c1_stuck")
    ring = torch.from_numpy(np.random.default_rng(5).standard_normal((3, int(fx["T"]), 2, int(fx["K"]))).astype(np.float32)).cuda()
    st = torch.tensor(fx["state_0"], device="cuda")
    torch.cuda.synchronize()
    out = []
    for use_n in (True, False):
        with native_planner_for(fx) as pl:
            pl.set_map(fx["R"]); pl.set_goal(fx["goal"])
            if use_n:
                pl.solve_n_async_device(7, st.data_ptr(), ring.data_ptr(), _capi.BN_NOISE_DEVICE_T2K, 3, ring[0].numel())
            else:
                for i in range(7):
                    pl.solve_async_device(st.data_ptr(), ring[i % 3].data_ptr(), _capi.BN_NOISE_DEVICE_T2K)
            pl.sync()
            out.append((pl.states(), pl.weights(), pl.get_mean()))
    for a, b in zip(*out):
        assert np.array_equal(a, b)


def test_zero_sigma_gives_uniform_weights_and_mean_controls():
    from benchnav_amd import NativeMPPI
    K, T, G = 256, 12, 32
    R = np.full((G, G), 0.2, np.float32)
    mean = np.tile(np.array([[0.7, -0.3]], np.float32), (T, 1))
    with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, sigmas=(0.0, 0.0), inv_var=(1.0, 1.0)) as pl:
        pl.set_map(R); pl.set_goal([12.0, 12.0]); pl.set_mean(mean)
        us, _ = pl.solve([4.0, 4.0, 0.0], np.random.default_rng(0).standard_normal((K, T, 2)).astype(np.float32))
        assert np.allclose(pl.weights(), 1.0 / K, rtol=1e-6)
        assert np.allclose(us[0], mean, atol=1e-6)
        X = pl.states()
        assert np.array_equal(X[0], X[-1])


def test_batched_instances_equal_single_instance_solves():
    O = _oracle()
    from benchnav_amd import NativeMPPI
    B, K, T, G = 3, 192, 16, 48
    rng = np.random.default_rng(7)
    maps = (rng.random((B, G, G)) * 0.9).astype(np.float32)
    states = np.stack([[5.0 + b, 6.0, 0.1 * b] for b in range(B)]).astype(np.float32)
    goals = np.stack([[20.0, 18.0 - b] for b in range(B)]).astype(np.float32)
    eps = rng.standard_normal((B, K, T, 2)).astype(np.float32)
    with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, num_instances=B, store_controls=True) as pl:
        for b in range(B):
            pl.set_map(maps[b], b); pl.set_goal(goals[b], b)
        us, xs = pl.solve(states, eps)
        for b in range(B):
            p = O.make_params(K, T, G, 0.5, goals[b], trig=O.TRIG_SPEC)
            orc = O.solve(p, maps[b], states[b], np.zeros((T, 2), np.float32), eps[b])
            assert_oracle_parity(oracle_metrics(native_outputs(pl, us, xs, b), orc), ctx=f"instance {b}")


def test_philox_noise_stream_is_what_the_kernel_consumes_and_is_normal():
    O = _oracle()
    fx = load_case("c1_basic")
    p = oracle_params_for(fx, O.TRIG_SPEC)
    with native_planner_for(fx) as pl:
        pl.set_map(fx["R"]); pl.set_goal(fx["goal"])
        draws = []
        for i in range(3):
            mean = pl.get_mean()
            us, xs = pl.solve(fx["state_0"])                   # eps=None -> in-kernel Philox
            eps = pl.philox_noise(i)
            draws.append(eps)
            orc = O.solve(p, fx["R"], fx["state_0"], mean, eps)
            assert_oracle_parity(oracle_metrics(native_outputs(pl, us, xs), orc), ctx=f"philox solve {i}")
        assert not np.array_equal(draws[0], draws[1])
    with native_planner_for(load_case("c2")) as pl:
        e = pl.philox_noise(5).astype(np.float64).ravel()       # 102400 draws
        assert abs(e.mean()) < 0.02 and abs(e.std() - 1.0) < 0.02
        assert abs((e ** 4).mean() - 3.0) < 0.15 and abs((e ** 3).mean()) < 0.05
        assert abs(np.corrcoef(e[0::2], e[1::2])[0, 1]) < 0.02


def test_top_samples_match_a_host_sort():
    fx = load_case("cvar")
    with native_planner_for(fx) as pl:
        pl.set_map(fx["R"]); pl.set_goal(fx["goal"])
        pl.solve(fx["state_0"], fx["eps_0"])
        w, X = pl.weights(), pl.states()
        n = 40
        s, tw = pl.top_samples(n)
        order = np.lexsort((np.arange(w.size), -w))[:n]
        assert np.array_equal(tw, w[order]) and np.all(np.diff(tw) <= 0)
        pos = tw > 0                                            # ties among zero weights are order-free
        assert np.array_equal(s[pos], X[order][pos])
        with pytest.raises(Exception):
            pl.top_samples(int(fx["K"]) + 1)                   # mppi.py:229 asserts n <= K


def test_errors_are_loud():
    from benchnav_amd import NativeMPPI
    from benchnav_amd._capi import BenchnavError
    with NativeMPPI(horizon=4, num_samples=64, grid_size=8, resolution=1.0) as pl:
        with pytest.raises(BenchnavError, match="set_map"):
            pl.solve([1.0, 1.0, 0.0])
        with pytest.raises(BenchnavError):
            pl.controls()                                       # not stored without the flag
    with pytest.raises(BenchnavError, match="LDS"):
        NativeMPPI(horizon=40000, num_samples=64, grid_size=8, resolution=1.0)      # (horizon=400 takes the slow path: test_gpu_census.py)


@pytest.mark.parametrize("K,T,B,sampled", [(4096, 50, 1, False), (2112, 33, 2, False), (16384, 100, 1, False), (8192, 50, 1, True), (2048, 20, 3, True)],
                         ids=["K4096", "K2112-B2", "c5", "sampled-c3", "sampled-B3"])
def test_one_launch_ticket_chain_equals_two_launch_chain(K, T, B, sampled):
    """K > 2048 (and every sampled-slip solve): the last workgroup of a launch merges the partials (ticket) and the
    tail of solve i rides in the launch of solve i+1.  A chain of warm-started Philox solves must give bit-identical
    results to the two-launch path (BN_FLAG_NO_PIPELINE), including X* and the weights of the last solve."""
    import torch
    from benchnav_amd import NativeMPPI, synth
    G = 512 if K > 8192 else 256
    inst = synth.make_instance(G, seed=6)
    sg = synth.slip_std_map(G, seed=6).numpy()
    n = 4
    states = [torch.from_numpy(np.tile(inst.start.numpy(), (B, 1)) + 0.2 * i + 0.1 * np.arange(B)[:, None]).float().cuda() for i in range(n)]
    torch.cuda.synchronize()
    results = {}
    for mode in ("one_launch", "flushed", "two_launch"):
        with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, num_instances=B, shared_map=True, seed=5,
                        sampled_slip=sampled, pipeline=(mode != "two_launch")) as pl:
            pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
            if sampled:
                pl.set_slip_std(sg)
            for i in range(n):
                pl.solve_async_device(states[i].data_ptr())
                if mode == "flushed":
                    pl.sync()
            pl.sync()
            xs = torch.as_tensor(_dev(pl, T, B), device="cuda").cpu().numpy()
            results[mode] = [(pl.states(b), pl.costs(b), pl.weights(b), pl.get_mean(b), xs[b].copy()) for b in range(B)]
    for mode in ("flushed", "two_launch"):
        for b in range(B):
            for j, (got, ref) in enumerate(zip(results["one_launch"][b], results[mode][b])):
                assert np.array_equal(got, ref), (mode, b, j)
    assert np.isfinite(results["one_launch"][0][4]).all() and np.abs(results["one_launch"][0][3]).max() > 0


def _dev(pl, T, B):
    from benchnav_amd import _capi
    from benchnav_amd.mppi import _DevArray
    return _DevArray(pl.device_buffer(_capi.BN_BUF_XSTAR)[0], (B, T + 1, 3))


@pytest.mark.parametrize("K,T,B,sampled", [(8192, 50, 2, False), (8192, 50, 1, True), (2112, 20, 7, False)], ids=["K8192-B2", "sampled", "B7"])
def test_ticket_handoff_stress(K, T, B, sampled):
    """The in-launch hand-off of the partials (sc1 stores -> ticket -> sc1 loads, possibly across XCDs) under load:
    300 back-to-back warm-started solves must end exactly where the two-launch chain ends; one stale word anywhere
    would show in the final mean (every solve's U* feeds the next)."""
    import torch
    from benchnav_amd import NativeMPPI, synth
    G, n = 256, 300
    inst = synth.make_instance(G, seed=8)
    st = torch.from_numpy(np.tile(inst.start.numpy(), (B, 1)) + 0.1 * np.arange(B)[:, None]).float().cuda()
    torch.cuda.synchronize()
    got = {}
    for mode in ("one_launch", "two_launch"):
        with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, num_instances=B, shared_map=True, seed=11,
                        sampled_slip=sampled, pipeline=(mode == "one_launch")) as pl:
            pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
            if sampled:
                pl.set_slip_std(synth.slip_std_map(G, seed=8).numpy())
            pl.solve_n_async_device(n, st.data_ptr())
            pl.sync()
            got[mode] = [(pl.get_mean(b), pl.costs(b), pl.weights(b)) for b in range(B)]
    for b in range(B):
        for a_, b_ in zip(got["one_launch"][b], got["two_launch"][b]):
            assert np.array_equal(a_, b_), b


@pytest.mark.parametrize("res,G", [(0.3, 50), (0.1, 200), (0.7, 40), (1.7, 30), (0.45, 64)])
def test_general_resolution_takes_the_validated_quotient(res, G):
    """A resolution that is not a power of two: the in-loop cell index divides with the three-instruction correctly rounded
    quotient, validated on the device over every float the lookups can see when the handle is created (bn_mppi_fast_quotient == 1),
    and the solve stays bit-exact against the oracle, whose lookup is the reference's true division (grid_map.py:203)."""
    from benchnav_amd import NativeMPPI, synth
    from oracle import oracle as O
    K, T = 512, 40
    rng = np.random.default_rng(int(res * 100))
    R = synth.iid_risk_map(G, 3).numpy()
    ext = G * res
    state = np.array([0.31 * ext, 0.43 * ext, -0.4], np.float32)
    goal = np.array([0.7 * ext, 0.6 * ext], np.float32)
    eps = rng.standard_normal((K, T, 2)).astype(np.float32)
    mean = np.clip(rng.standard_normal((T, 2)) * 0.2 + [0.7, 0.0], [0, -1], [1, 1]).astype(np.float32)
    orc = O.solve(O.make_params(K, T, G, res, goal, trig=O.TRIG_SPEC), R, state, mean, eps)
    for kernel in ("auto", "role", "wave"):
        with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=res, store_controls=True, kernel=kernel) as pl:
            assert pl.fast_quotient() == 1
            pl.set_map(R); pl.set_goal(goal); pl.set_mean(mean)
            us, xs = pl.solve(state, eps)
            assert_oracle_parity(oracle_metrics(native_outputs(pl, us, xs), orc), ctx=f"res={res} {kernel}")
    with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5) as pl:
        assert pl.fast_quotient() == 2
