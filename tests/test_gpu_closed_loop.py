"""GPU: device-side closed loop ("next" row N2) -- PlanetaryEnv.step between consecutive solves
(planetary_env.py:189-219 in observation mode, traversability_model.py:65-69).

Every logged transition is checked against the oracle's env step on the logged inputs (bit-exact), every
solve of the episode against the oracle's solve on the logged state with the logged warm start, and the
whole pipelined episode against the same episode driven step by step from the host."""
import numpy as np
import pytest

from helpers import assert_oracle_parity, load_case, native_outputs, native_planner_for, oracle_metrics, oracle_params_for

pytestmark = pytest.mark.gpu


def _latent(G, seed):
    from benchnav_amd import synth
    return (synth.smooth_risk_map(G, seed) * 0.6).numpy(), synth.slip_std_map(G, seed).numpy()


def _cell(p, v, origin):
    return int(np.clip(np.floor(np.float32(np.float32(v) - np.float32(origin)) / np.float32(p.res)), 0, p.G - 1))


@pytest.mark.parametrize("arith", ["spec", "reference_order"])
def test_episode_transitions_and_solves_match_the_oracle(arith):
    import torch
    from oracle import oracle as O
    from benchnav_amd import _capi
    fx = load_case("c1_basic")
    K, T, G = int(fx["K"]), int(fx["T"]), int(fx["G"])
    n = 6
    lat_mean, lat_std = _latent(G, 9)
    rng = np.random.default_rng(2)
    z = rng.standard_normal((n, 1)).astype(np.float32)
    eps = rng.standard_normal((n, K, T, 2)).astype(np.float32)
    zd, ed = torch.from_numpy(z).cuda(), torch.from_numpy(eps).cuda()
    torch.cuda.synchronize()
    p = oracle_params_for(fx, O.TRIG_SPEC if arith == "spec" else O.TRIG_SPEC_PER_STEP)

    def run(steps):
        with native_planner_for(fx, reference_order=(arith == "reference_order")) as pl:
            assert pl.arithmetic() == arith
            pl.set_map(fx["R"]); pl.set_goal(fx["goal"])
            pl.env_attach(lat_mean, lat_std, goal_threshold=1.0, delta_t=0.1)
            log = pl.episode(steps, fx["state_0"], z_device_ptr=zd.data_ptr(), eps_ptr=ed.data_ptr(),
                             kind=_capi.BN_NOISE_DEVICE_KT2, eps_ring=n, eps_stride=eps[0].size)
            mean = pl.get_mean()
            return log, pl.last_actions, native_outputs(pl, mean[None], np.zeros((1, T + 1, 3), np.float32))

    (states, rewards, done), actions, last = run(n)
    (_, _, _), _, before_last = run(n - 1)           # its final mean = the warm start of the n-step episode's last solve
    assert states.shape == (n + 1, 1, 3) and np.array_equal(states[0, 0], fx["state_0"]) and done[0] == -1
    for i in range(n):
        # env step i (planetary_env.py:203-205): slip sampled at the cell of state i, trav = 1 - clamp(slip, 0, 1),
        # observation-mode transit of the applied control; bit-exact against the oracle's env step
        ix, iy = _cell(p, states[i, 0, 0], p.x0), _cell(p, states[i, 0, 1], p.y0)
        slip = np.float32(np.float32(z[i, 0] * lat_std[iy, ix]) + lat_mean[iy, ix])
        trav = np.float32(1.0) - np.clip(slip, np.float32(0), np.float32(1))
        assert rewards[i, 0] == trav
        assert np.array_equal(O.env_step(p, float(trav), states[i, 0], actions[i, 0]), states[i + 1, 0])
    # the last solve of the pipelined episode, in full, against the oracle on the logged state with the planner's own warm start
    orc = O.solve(p, fx["R"], states[n - 1, 0], before_last["Ustar"], eps[n - 1])
    last["Ustar"] = last["Ustar"]                   # get_mean() == U* of the last solve
    assert_oracle_parity(oracle_metrics(last, dict(orc, Xstar=last["Xstar"])), ctx="last solve of the episode")
    assert np.abs(actions[n - 1, 0] - orc["Ustar"][0]).max() < 2e-6


def test_pipelined_episode_equals_host_driven_loop_bitwise():
    """The same closed loop driven from the host: solve (sync), read U*[0], apply the oracle-free device env step by
    running a 1-step episode... instead: compare against a second handle running the episode in two halves."""
    import torch
    from benchnav_amd import _capi
    fx = load_case("c1_stuck")
    G = int(fx["G"])
    lat_mean, lat_std = _latent(G, 5)
    n, B = 8, 3
    rng = np.random.default_rng(4)
    zd = torch.from_numpy(rng.standard_normal((n, B)).astype(np.float32)).cuda()
    starts = np.stack([fx["state_0"] + np.float32([0.5 * b, 0.2 * b, 0.1 * b]) for b in range(B)]).astype(np.float32)
    torch.cuda.synchronize()
    logs = []
    for split in (False, True):
        with native_planner_for(fx, num_instances=B, shared_map=True) as pl:
            pl.set_map(fx["R"]); pl.set_goal(fx["goal"])
            pl.env_attach(lat_mean, lat_std)
            if not split:
                logs.append(pl.episode(n, starts, z_device_ptr=zd.data_ptr()))
            else:
                s1, r1, d1 = pl.episode(n // 2, starts, z_device_ptr=zd.data_ptr())
                logs.append((s1, r1, d1))
    full, half = logs
    assert np.array_equal(full[0][: n // 2 + 1], half[0]) and np.array_equal(full[1][: n // 2], half[1])
    assert np.isfinite(full[0]).all() and (np.abs(np.diff(full[0][:, :, :2], axis=0)) <= 0.1 + 1e-6).all()


@pytest.mark.parametrize("freeze", [True, False], ids=["freeze-opt-in", "reference-default"])
def test_goal_arrival(freeze):
    fx = load_case("c1_basic")
    G = int(fx["G"])
    lat_mean, lat_std = np.zeros((G, G), np.float32), np.zeros((G, G), np.float32)     # slip 0: full traversability
    start = np.array([float(fx["goal"][0]) - 1.3, float(fx["goal"][1]), 0.0], np.float32)   # 1.3 m from the goal, heading to it
    with native_planner_for(fx) as pl:
        pl.set_map(np.zeros((G, G), np.float32)); pl.set_goal(fx["goal"])
        pl.env_attach(lat_mean, lat_std, goal_threshold=1.0, freeze_on_goal=freeze)
        states, rewards, done = pl.episode(30, start)
    d = np.linalg.norm(states[:, 0, :2] - fx["goal"][None], axis=1)
    assert done[0] >= 0, "the rover should reach the 1 m goal disc within 30 steps"
    k = done[0]
    assert d[k + 1] < 1.0 <= d[k]                                  # first state inside the disc is the one after step k
    if freeze:      # opt-in: an instance inside the goal disc stays put
        assert np.array_equal(states[k + 1:], np.repeat(states[k + 1:k + 2], len(states) - k - 1, axis=0))
    else:           # the reference's PlanetaryEnv keeps moving when a terminated episode is stepped (planetary_env.py:189-219)
        assert not np.array_equal(states[k + 1], states[k + 3])
    assert (rewards == 1.0).all()
