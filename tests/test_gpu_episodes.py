"""GPU: episode-level outcomes of the device-side closed loop against the reference's own closed loop
(SURVEY.md 8a parity spec (vi); driver of test_mppi.py:171-198; fixture tests/golden/episodes.npz from the REAL
MPPI + PlanetaryEnv).

(a) statistical: 32 free-running episodes (own Philox noise / slip draws) against the reference's 32 seeds -- all reach
    the goal, the steps-to-goal distribution and the mean traversability observed agree;
(b) like for like: ONE episode replayed with the reference's noise blocks and slip draws, free-running on the device:
    the trajectory stays within tolerance of the reference's and arrives at the same control step."""
import numpy as np
import pytest

from helpers import load_case

pytestmark = pytest.mark.gpu


def _start_state(fx):
    d = fx["goal"] - fx["start"]
    return np.array([fx["start"][0], fx["start"][1], np.arctan2(d[1], d[0])], np.float32)      # planetary_env.py:128-141


def test_outcome_distribution_matches_the_reference_over_32_seeds():
    from benchnav_amd import NativeMPPI
    fx = load_case("episodes")
    B, K, T, G = int(fx["n_seeds"]), int(fx["K"]), int(fx["T"]), int(fx["G"])
    n = 260
    with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=float(fx["res"]), num_instances=B, shared_map=True,
                    stuck_threshold=float(fx["thr"]), seed=2024) as pl:
        pl.set_map(fx["R"]); pl.set_goal(fx["goal"])
        pl.env_attach(fx["MU"], fx["SG"], goal_threshold=float(fx["goal_threshold"]), delta_t=float(fx["delta_t"]), seed=7, freeze_on_goal=True)
        states, rewards, done = pl.episode(n, np.tile(_start_state(fx), (B, 1)))
    assert (done >= 0).all(), f"reference: {int(fx['reached'].sum())}/{B} reach the goal; here {(done >= 0).sum()}/{B}"
    steps = done + 1
    ref = fx["steps"]
    assert abs(np.median(steps) - np.median(ref)) <= 4, (np.median(steps), np.median(ref))
    assert steps.min() >= ref.min() - 8 and steps.max() <= ref.max() + 8, (steps.min(), steps.max(), ref.min(), ref.max())
    final = np.array([np.linalg.norm(states[done[b] + 1, b, :2] - fx["goal"]) for b in range(B)])
    assert (final < float(fx["goal_threshold"])).all() and abs(final.mean() - fx["final_dist"].mean()) <= 0.03
    mean_reward = np.array([rewards[:done[b] + 1, b].mean() for b in range(B)])
    assert abs(mean_reward.mean() - fx["mean_reward"].mean()) <= 0.02, (mean_reward.mean(), fx["mean_reward"].mean())


def test_replayed_reference_episode_free_running():
    import torch
    from benchnav_amd import NativeMPPI, _capi
    fx = load_case("episodes")
    K, T, G = int(fx["ep_K"]), int(fx["ep_T"]), int(fx["G"])
    n = len(fx["ep_z"])
    eps = torch.from_numpy(fx["ep_eps"]).cuda()                       # (n, K, T, 2): block i is the noise of solve i
    z = torch.from_numpy(fx["ep_z"]).cuda()
    with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=float(fx["res"]), stuck_threshold=float(fx["thr"])) as pl:
        pl.set_map(fx["R"]); pl.set_goal(fx["goal"])
        pl.env_attach(fx["MU"], fx["SG"], goal_threshold=float(fx["goal_threshold"]), delta_t=float(fx["delta_t"]))
        states, rewards, done = pl.episode(n, fx["ep_states"][0], z_device_ptr=z.data_ptr(), eps_ptr=eps.data_ptr(),
                                           kind=_capi.BN_NOISE_DEVICE_KT2, eps_ring=n, eps_stride=K * T * 2)
        actions = pl.last_actions
    assert np.abs(states[:, 0] - fx["ep_states"]).max() <= 2e-3, np.abs(states[:, 0] - fx["ep_states"]).max()
    assert np.abs(actions[:, 0] - fx["ep_actions"]).max() <= 2e-2
    first = int(np.argmax(fx["ep_terminated"]))
    assert fx["ep_terminated"][first] and done[0] == first            # arrives at the very same control step


@pytest.mark.parametrize("reference_order", [False, True])
def test_reference_episode_at_baseline_size(reference_order):
    """the closed loop at BASELINE configs[1] size (K=1024, T=50, 256x256 map) against 75 control steps of the REAL
    MPPI + PlanetaryEnv (tests/golden/episode_c2.npz; the noise blocks are regenerated from stored generator states), in both
    arithmetics.  (1) teacher-forced through the C ABI -- the reference's state and previous U* in front of every solve: every U*
    within SURVEY 8a (iv).  (2) free-running on the device (bn_mppi_episode_async, the reference's noise blocks and slip draws): a
    closed loop that puts most of a solve's weight on one rollout is chaotic, the fixture holds the reference's own run from a
    starts one or two ulps away, and the device is held to what those runs span: their range of arrival steps (+-2) and twice their
    largest deviation (tests/test_episodes_golden.py has the oracle's side of the same check)."""
    import torch
    from benchnav_amd import NativeMPPI, _capi
    from helpers import episode_c2_bounds, regenerate_noise_blocks
    fx = load_case("episode_c2")
    eps_h = regenerate_noise_blocks(fx)
    K, T, G, n = int(fx["K"]), int(fx["T"]), int(fx["G"]), len(fx["ep_z"])
    common = dict(horizon=T, num_samples=K, grid_size=G, resolution=float(fx["res"]), stuck_threshold=float(fx["thr"]), reference_order=reference_order)
    with NativeMPPI(**common) as pl:
        assert pl.arithmetic() == ("reference_order" if reference_order else "spec")
        pl.set_map(fx["R"]); pl.set_goal(fx["goal"])
        worst = 0.0
        for i in range(n):
            pl.set_mean(fx["ep_ustar"][i - 1] if i else None)
            us, xs = pl.solve(fx["ep_states"][i], eps_h[i])
            worst = max(worst, float(np.abs(us[0] - fx["ep_ustar"][i]).max()))
        assert worst <= 2e-3, worst                                    # 2e-2; the oracle reaches 6e-5
    lo, hi, dev_max = episode_c2_bounds(fx)
    eps = torch.from_numpy(eps_h).cuda()
    z = torch.from_numpy(np.resize(fx["ep_z"], hi)).cuda()           # (beyond the stored episode draws and blocks repeat, as in the fixture's perturbed runs)
    with NativeMPPI(**common) as pl:
        pl.set_map(fx["R"]); pl.set_goal(fx["goal"])
        pl.env_attach(fx["MU"], fx["SG"], goal_threshold=float(fx["goal_threshold"]), delta_t=float(fx["delta_t"]))
        states, rewards, done = pl.episode(hi, fx["ep_states"][0], z_device_ptr=z.data_ptr(), eps_ptr=eps.data_ptr(),
                                           kind=_capi.BN_NOISE_DEVICE_KT2, eps_ring=n, eps_stride=K * T * 2)
    assert done[0] >= 0 and lo <= int(done[0]) + 1 <= hi, (done[0], lo, hi)
    m = min(int(done[0]) + 2, len(fx["ep_states"]))
    dev = np.abs(states[:m, 0] - fx["ep_states"][:m]).max()
    assert dev <= dev_max, (dev, dev_max)
