"""CPU: the oracle driven free-running through the reference's closed loop (test_mppi.py:171-198: solve -> env.step ->
...) on the noise blocks and slip draws of a REAL reference episode (fixture tests/golden/episodes.npz, ep_* arrays:
reference MPPI + reference PlanetaryEnv, K=64, T=12).  Free-running: every U* feeds the next solve's mean and the next
state, so per-solve differences (libm / spec sin-cos, summation order) may compound; the bar is the trajectory tolerance
of SURVEY.md 8a scaled by the episode length, and the same arrival step."""
import numpy as np

from helpers import load_case
from oracle import oracle as O


def _replay(fx, trig):
    K, T, G = int(fx["ep_K"]), int(fx["ep_T"]), int(fx["G"])
    p = O.make_params(K, T, G, float(fx["res"]), fx["goal"], thr=float(fx["thr"]), trig=trig)
    s = fx["ep_states"][0].copy()
    mean = np.zeros((T, 2), np.float32)
    states, acts, term = [s.copy()], [], []
    for i in range(len(fx["ep_z"])):
        out = O.solve(p, fx["R"], s, mean, fx["ep_eps"][i])
        mean = out["Ustar"]
        s, rw, t = O.env_step_sampled(p, fx["MU"], fx["SG"], float(fx["ep_z"][i]), float(fx["goal_threshold"]), s, mean[0])
        states.append(s.copy()); acts.append(mean[0].copy()); term.append(t)
    return np.asarray(states), np.asarray(acts), np.asarray(term)


def test_oracle_free_running_episode_follows_the_reference():
    fx = load_case("episodes")
    for trig in (O.TRIG_LIBM, O.TRIG_SPEC):
        states, acts, term = _replay(fx, trig)
        assert np.abs(states - fx["ep_states"]).max() <= 2e-3, (trig, np.abs(states - fx["ep_states"]).max())
        assert np.abs(acts - fx["ep_actions"]).max() <= 2e-2
        assert term.tolist() == fx["ep_terminated"].tolist()
    assert fx["ep_terminated"][-1], "the stored episode ends with the goal reached"


def test_reference_outcome_statistics_are_what_the_gpu_test_expects():
    fx = load_case("episodes")
    assert int(fx["n_seeds"]) >= 32 and fx["reached"].all()
    assert 150 <= np.median(fx["steps"]) <= 200 and (fx["final_dist"] < float(fx["goal_threshold"])).all()
