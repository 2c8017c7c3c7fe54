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
