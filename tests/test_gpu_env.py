"""GPU: the PlanetaryEnv mirror for B environments (reset / step / collision_check, planetary_env.py:143-232):
return shapes of the reference with a leading batch dimension, every transition bit-exact against the oracle's
env step on the same slip draws, collision flags against the NumPy statement of the observation-mode lookup."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(B=5, G=64, seed=3, K=128, T=12):
    from benchnav_amd import NativeMPPI, synth
    from benchnav_amd.env import BatchedPlanetaryEnv
    mu = (synth.smooth_risk_map(G, seed) * 0.5).numpy()
    sg = synth.slip_std_map(G, seed).numpy()
    rng = np.random.default_rng(seed)
    ext = G * 0.5
    start = rng.uniform(0.2 * ext, 0.4 * ext, (B, 2)).astype(np.float32)
    goal = rng.uniform(0.6 * ext, 0.8 * ext, (B, 2)).astype(np.float32)
    pl = NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, num_instances=B, shared_map=True, stream=0)
    pl.set_map(mu)
    env = BatchedPlanetaryEnv(pl, mu, sg, start, goal, stuck_threshold=0.05, goal_threshold=1.0, seed=4)
    return pl, env, mu, sg, start, goal


def _cell(v, G):
    return int(np.clip(np.floor(np.float32(v) / np.float32(0.5)), 0, G - 1))


def test_reset_step_collision_check_shapes_and_values():
    import torch
    from oracle import oracle as O
    pl, env, mu, sg, start, goal = _setup()
    B, G = env.B, 64
    s0 = env.reset(seed=11)
    assert s0.shape == (B, 3) and s0.is_cuda
    assert np.allclose(s0[:, :2].cpu().numpy(), start) and np.allclose(s0[:, 2].cpu().numpy(), np.arctan2(goal[:, 1] - start[:, 1], goal[:, 0] - start[:, 0]), atol=1e-6)
    assert torch.isnan(env._reward).all()
    rng = np.random.default_rng(0)
    state = s0.cpu().numpy().copy()
    p = O.make_params(128, 12, G, 0.5, goal[0], trig=O.TRIG_SPEC)
    for i in range(8):
        a = np.stack([rng.uniform(0, 1, B), rng.uniform(-1, 1, B)], 1).astype(np.float32)
        z = rng.standard_normal(B).astype(np.float32)
        ns, reward, term, trunc = env.step(torch.from_numpy(a).cuda(), z=torch.from_numpy(z).cuda())
        assert ns.shape == (B, 3) and reward.shape == (B,) and term.shape == (B,) and term.dtype == torch.bool and trunc is False
        ns_h, rw_h = ns.cpu().numpy(), reward.cpu().numpy()
        for b in range(B):
            ix, iy = _cell(state[b, 0], G), _cell(state[b, 1], G)
            slip = np.float32(np.float32(z[b] * sg[iy, ix]) + mu[iy, ix])
            trav = np.float32(1.0) - np.clip(slip, np.float32(0), np.float32(1))
            assert rw_h[b] == trav
            assert np.array_equal(O.env_step(p, float(trav), state[b], a[b]), ns_h[b]), (i, b)
            assert bool(term[b]) == bool(np.linalg.norm(ns_h[b, :2] - goal[b]) < 1.0)
        state = ns_h.copy()
    # collision_check: (B, N, 3) -> (B, N) bool with one draw per position
    N = 7
    pos = np.concatenate([rng.uniform(0, 32, (B, N, 2)), np.zeros((B, N, 1))], 2).astype(np.float32)
    zc = rng.standard_normal((B, N)).astype(np.float32)
    got = env.collision_check(torch.from_numpy(pos).cuda(), z=torch.from_numpy(zc).cuda())
    assert got.shape == (B, N) and got.dtype == torch.bool
    want = np.zeros((B, N), bool)
    for b in range(B):
        for n in range(N):
            ix, iy = _cell(pos[b, n, 0], G), _cell(pos[b, n, 1], G)
            slip = np.float32(np.float32(zc[b, n] * sg[iy, ix]) + mu[iy, ix])
            want[b, n] = (np.float32(1.0) - np.clip(slip, np.float32(0), np.float32(1))) <= np.float32(0.05)
    assert np.array_equal(got.cpu().numpy(), want)
    pl.close()


def test_philox_draws_are_seeded_and_truncation_follows_the_time_limit():
    import torch
    from benchnav_amd.env import BatchedPlanetaryEnv
    pl, env, mu, sg, start, goal = _setup(B=3)
    a = torch.tensor([[0.8, 0.1]] * 3, device="cuda")

    def episode(seed, n=5):
        env.reset(seed=seed)
        out = []
        for _ in range(n):
            s, r, t, tr = env.step(a)
            out.append((s.cpu().numpy().copy(), r.cpu().numpy().copy()))
        return out
    e1, e2, e3 = episode(5), episode(5), episode(6)
    assert all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) for x, y in zip(e1, e2))
    assert any(not np.array_equal(x[1], y[1]) for x, y in zip(e1, e3))
    r = np.stack([x[1] for x in e1])
    assert ((r >= 0) & (r <= 1)).all() and r.std() > 0
    env2 = BatchedPlanetaryEnv(pl, mu, sg, start, goal, stuck_threshold=0.05, time_limit=0.25, seed=1)
    flags = [env2.step(a)[3] for _ in range(4)]
    assert flags == [False, False, True, True]                     # elapsed 0.1, 0.2, 0.3 > 0.25 (planetary_env.py:218)
    c1 = env2.collision_check(torch.rand(3, 50, 3, device="cuda") * 30)
    c2 = env2.collision_check(torch.rand(3, 50, 3, device="cuda") * 30)
    assert c1.shape == (3, 50) and c1.dtype == torch.bool and c2.shape == (3, 50)
    pl.close()


def test_untraversable_start_is_refused_like_the_reference():
    from benchnav_amd import NativeMPPI
    from benchnav_amd.env import BatchedPlanetaryEnv
    G = 32
    mu = np.full((G, G), 0.99, np.float32); sg = np.zeros((G, G), np.float32)
    with NativeMPPI(horizon=5, num_samples=64, grid_size=G, resolution=0.5, stream=0) as pl:
        pl.set_map(mu)
        with pytest.raises(ValueError, match="not traversable"):
            BatchedPlanetaryEnv(pl, mu, sg, [4.0, 4.0], [10.0, 10.0], stuck_threshold=0.1)


def test_planner_in_the_loop_reaches_the_goal_and_fused_run_agrees_in_shape():
    """The reference's driver loop (test_mppi.py:171-198) with the batched mirror: solve -> step -> collision_check."""
    import torch
    from benchnav_amd import _capi
    from benchnav_amd.mppi import _DevArray
    pl, env, mu, sg, start, goal = _setup(B=4, K=512, T=20)
    B, T = env.B, 20
    us = torch.as_tensor(_DevArray(pl.device_buffer(_capi.BN_BUF_USTAR)[0], (B, T, 2)), device="cuda")
    xs = torch.as_tensor(_DevArray(pl.device_buffer(_capi.BN_BUF_XSTAR)[0], (B, T + 1, 3)), device="cuda")
    state = env.reset(seed=2)
    d0 = (state[:, :2] - env._goal_pos).norm(dim=1)
    reached = torch.zeros(B, dtype=torch.bool, device="cuda")
    for i in range(400):
        pl.solve_async_device(state.data_ptr()); pl.flush()
        state, reward, term, trunc = env.step(us[:, 0, :])
        coll = env.collision_check(xs)
        assert coll.shape == (B, T + 1)
        reached |= term
        if bool(reached.all()) or trunc:
            break
    d1 = (state[:, :2] - env._goal_pos).norm(dim=1)
    assert bool((d1 < d0).all()) and int(reached.sum()) >= 3, (d0, d1, reached)
    env.reset(seed=2)
    states, rewards, done = env.run(50)
    assert states.shape == (51, B, 3) and rewards.shape == (50, B) and done.shape == (B,)
    pl.close()


# ---- against the REAL PlanetaryEnv: fixture tests/golden/env.npz (make_golden.py run_env_case) --------------------------
def _fixture_env(B, fx, start=None, stuck_threshold=None):
    import torch
    from benchnav_amd import NativeMPPI
    from benchnav_amd.env import BatchedPlanetaryEnv
    G = int(fx["G"])
    pl = NativeMPPI(horizon=8, num_samples=64, grid_size=G, resolution=float(fx["res"]), num_instances=B, shared_map=True,
                    x_limits=fx["x_limits"].tolist(), y_limits=fx["y_limits"].tolist(), stream=torch.cuda.current_stream().cuda_stream)
    pl.set_map(fx["MU"])
    env = BatchedPlanetaryEnv(pl, fx["MU"], fx["SG"], fx["start"] if start is None else start, fx["goal"], delta_t=float(fx["delta_t"]),
                              time_limit=float(fx["time_limit"]), stuck_threshold=float(fx["stuck_threshold"]) if stuck_threshold is None else stuck_threshold,
                              goal_threshold=float(fx["goal_threshold"]), seed=1)
    return pl, env


def test_env_step_against_the_reference_environment_teacher_forced():
    """All 260 recorded transitions at once (one environment per recorded step, each started from the reference's state):
    bit-exact against the oracle, <= 1e-6 against the reference (sin/cos of SLEEF vs the kernel's spec), rewards exact."""
    import torch
    from helpers import load_case
    from oracle import oracle as O
    fx = load_case("env")
    n = len(fx["z"])
    pl, env = _fixture_env(n, fx)
    env._robot_state = torch.from_numpy(fx["states"][:n].copy()).cuda()
    ns, rw, term, trunc = env.step(torch.from_numpy(fx["actions"]).cuda(), z=torch.from_numpy(fx["z"]).cuda())
    ns, rw, term = ns.cpu().numpy(), rw.cpu().numpy(), term.cpu().numpy()
    assert np.array_equal(rw, fx["rewards"])
    assert np.abs(ns - fx["states"][1:]).max() <= 1e-6
    p = O.make_params(1, 1, int(fx["G"]), float(fx["res"]), fx["goal"], dt=float(fx["delta_t"]), x_limits=tuple(fx["x_limits"].tolist()),
                      y_limits=tuple(fx["y_limits"].tolist()), trig=O.TRIG_SPEC)
    for i in range(n):
        o_next, o_rw, o_term = O.env_step_sampled(p, fx["MU"], fx["SG"], float(fx["z"][i]), float(fx["goal_threshold"]), fx["states"][i], fx["actions"][i])
        assert np.array_equal(ns[i], o_next) and rw[i] == o_rw and bool(term[i]) == o_term, i
    edge = np.abs(np.linalg.norm(fx["states"][1:, :2] - fx["goal"], axis=1) - float(fx["goal_threshold"])) < 1e-5
    assert np.array_equal(term[~edge], fx["terminated"][~edge])
    pl.close()


def test_env_free_running_follows_the_reference_episode_and_does_not_freeze():
    import torch
    from helpers import load_case
    fx = load_case("env")
    pl, env = _fixture_env(1, fx)
    s = env.reset()
    assert np.abs(s.cpu().numpy()[0] - fx["states"][0]).max() <= 1e-6          # start position and atan2 heading (planetary_env.py:128-141)
    env._robot_state = torch.from_numpy(fx["states"][:1].copy()).cuda()
    a_all, z_all = torch.from_numpy(fx["actions"]).cuda(), torch.from_numpy(fx["z"]).cuda()
    worst, terms, truncs, states = 0.0, [], [], []
    for i in range(len(fx["z"])):
        s, rw, term, trunc = env.step(a_all[i:i + 1], z=z_all[i:i + 1])
        states.append(s)
        terms.append(term); truncs.append(trunc)
    got = torch.cat(states).cpu().numpy()
    assert np.abs(got - fx["states"][1:]).max() <= 1e-4
    terms = torch.cat(terms).cpu().numpy()
    first = int(np.argmax(fx["terminated"]))
    assert terms[first] and not terms[:first].any()
    assert not np.array_equal(got[first], got[first + 2])                     # stepped past the goal like the reference
    assert truncs == fx["truncated"].tolist()                                 # elapsed_time > time_limit (planetary_env.py:212,218)
    pl.close()


def test_collision_check_against_the_reference_environment():
    import torch
    from helpers import load_case
    fx = load_case("env")
    B, N = fx["cc_z"].shape
    pl, env = _fixture_env(B, fx, start=np.tile(fx["start"], (B, 1)))
    pos = torch.from_numpy(fx["cc_states"]).cuda()
    got = env.collision_check(pos, z=torch.from_numpy(fx["cc_z"]).cuda())
    assert got.shape == (B, N) and np.array_equal(got.cpu().numpy(), fx["cc_out"])
    env.stuck_threshold = float(fx["cc2_threshold"])
    got2 = env.collision_check(pos, z=torch.from_numpy(fx["cc2_z"]).cuda())
    assert np.array_equal(got2.cpu().numpy(), fx["cc2_out"])
    pl.close()


def test_a_private_planner_stream_is_refused():
    """the environment's kernels and torch's tensors must share a stream."""
    from benchnav_amd import NativeMPPI
    from benchnav_amd.env import BatchedPlanetaryEnv
    G = 32
    mu = np.full((G, G), 0.1, np.float32); sg = np.zeros((G, G), np.float32)
    with NativeMPPI(horizon=5, num_samples=64, grid_size=G, resolution=0.5) as pl:      # stream=None: private stream
        pl.set_map(mu)
        with pytest.raises(RuntimeError, match="current stream"):
            BatchedPlanetaryEnv(pl, mu, sg, [4.0, 4.0], [10.0, 10.0])


def test_from_reference_shaped_environment():
    """BatchedPlanetaryEnv.from_reference on an object carrying exactly the attributes the REAL PlanetaryEnv had when
    tests/golden/make_golden.py ran (boundary.json) and the values it held (boundary.npz): same start state as the real
    environment's, thresholds and time step taken over."""
    import json
    import os
    import types
    import torch
    from benchnav_amd import NativeMPPI
    from benchnav_amd.env import BatchedPlanetaryEnv
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    meta = json.load(open(os.path.join(here, "boundary.json")))
    fx = np.load(os.path.join(here, "boundary.npz"))
    sc = fx["env_scalars"]
    G, res = int(sc[5]), float(sc[6])
    latent = torch.distributions.Normal(torch.as_tensor(fx["MU"]), torch.as_tensor(fx["SG"]))
    gm = types.SimpleNamespace(**{k: None for k in meta["attributes"]["GridMap"]})
    gm.grid_size, gm.resolution, gm.x_limits, gm.y_limits = G, res, (float(sc[7]), float(sc[8])), (float(sc[9]), float(sc[10]))
    gm.distributions = {"latent_models": latent}
    env_ref = types.SimpleNamespace(**{k: None for k in meta["attributes"]["PlanetaryEnv"]})
    env_ref._grid_map, env_ref._start_pos, env_ref._goal_pos = gm, torch.as_tensor(fx["env_start"]), torch.as_tensor(fx["env_goal"])
    env_ref._delta_t, env_ref._time_limit, env_ref.stuck_threshold, env_ref._goal_threshold, env_ref._seed = (float(sc[0]), float(sc[1]), float(sc[2]),
                                                                                                           float(sc[3]), int(sc[4]))
    B = 3
    with NativeMPPI(horizon=10, num_samples=64, grid_size=G, resolution=res, num_instances=B, shared_map=True, stream=0) as pl:
        pl.set_map(fx["MU"])
        env = BatchedPlanetaryEnv.from_reference(pl, env_ref)
        s0 = env.reset()
        assert s0.shape == (B, 3)
        assert np.allclose(s0.cpu().numpy(), np.broadcast_to(fx["env_robot_state0"], (B, 3)), atol=1e-6)
        assert env._delta_t == float(sc[0]) and env._time_limit == float(sc[1]) and env.stuck_threshold == float(sc[2]) and env._goal_threshold == float(sc[3])
        ns, reward, term, trunc = env.step(torch.tensor([[0.5, 0.1]] * B, device="cuda"))
        assert ns.shape == (B, 3) and bool(torch.isfinite(ns).all()) and trunc is False
