"""DWA on the MPPI transit/cost kernels ("next" row N3): reference src/planners/local_planners/dwa.py.
CPU: the C oracle against the fixture captured from the reference.  GPU: kernel vs oracle (bit-exact
trajectories and costs) and vs the fixture; the drop-in class end to end."""
import os

import numpy as np
import pytest

from helpers import GOLDEN_DIR
from oracle import oracle as O


def _fx():
    z = np.load(os.path.join(GOLDEN_DIR, "dwa.npz"))
    return {k: z[k] for k in z.files}


def _params(fx):
    return O.make_params(100, int(fx["T"]), int(fx["G"]), float(fx["res"]), fx["goal"], thr=float(fx["thr"]), trig=O.TRIG_SPEC)


def test_oracle_dwa_matches_reference_fixture():
    fx = _fx()
    p = _params(fx)
    for i in range(int(fx["n_solves"])):
        got = O.dwa(p, fx["R"], fx[f"state_{i}"], fx[f"actions_{i}"], fx[f"sub_goal_{i}"])
        assert np.abs(got["X"] - fx[f"X_{i}"]).max() <= 1e-4
        c = fx[f"cost_{i}"]
        assert (np.abs(got["cost"] - c) <= 1e-3 * np.maximum(1, np.abs(c))).all()
        assert got["best"] == int(np.argmin(c))
        assert np.abs(got["w"] - fx[f"w_{i}"]).max() <= 5e-3
        assert np.abs(got["X"][got["best"]] - fx[f"x_opt_{i}"]).max() <= 1e-4


def test_sub_goal_is_picked_from_candidate_zeros_aliased_slot():
    """the reference's sub-goal rule runs on state_seq_batch[0, 0, :] after the rollouts (dwa.py:240-244), i.e. on
    the input state advanced by candidate 0's first un-clamped step.  The fixture holds threshold-adjacent states where
    that changes the pick (sub_goal_naive = what the input state itself would give)."""
    fx = _fx()
    p = _params(fx)
    changed = 0
    for i in range(int(fx["n_solves"])):
        sg, sel, idx = O.dwa_sub_goal(p, fx["R"], fx[f"state_{i}"], fx[f"actions_{i}"][0], fx["path"], float(fx["lookahead"]))
        assert np.abs(sel - fx[f"sub_goal_state_{i}"]).max() <= 1e-6
        assert np.array_equal(sg, fx[f"sub_goal_{i}"]), i
        changed += int(not np.array_equal(fx[f"sub_goal_{i}"], fx[f"sub_goal_naive_{i}"]))
    assert changed >= 2


def test_drop_in_class_sub_goal_state_on_the_host():
    """benchnav_amd.DWA's host-side restatement of that slot-0 state and the rule on it (no GPU needed for this part)."""
    import torch
    from benchnav_amd.dwa import DWA
    fx = _fx()
    d = DWA.__new__(DWA)
    torch.nn.Module.__init__(d)
    d._dtype = torch.float32
    d._risk_cpu = torch.tensor(fx["R"])
    G, res = int(fx["G"]), float(fx["res"])
    d._grid = (G, res, (0.0, G * res), (0.0, G * res))
    d._u_min, d._u_max = torch.tensor([0.0, -1.0]), torch.tensor([1.0, 1.0])
    d._lookahead_distance = float(fx["lookahead"])
    d.reference_path = torch.tensor(fx["path"])
    same_build = str(fx["torch_version"]) == torch.__version__
    for i in range(int(fx["n_solves"])):
        sel = d._sub_goal_state(torch.tensor(fx[f"state_{i}"]), torch.tensor(fx[f"actions_{i}"][0]))
        assert np.abs(sel.numpy() - fx[f"sub_goal_state_{i}"]).max() <= (0 if same_build else 1e-6)
        assert np.array_equal(d._select_sub_goal(sel).numpy(), fx[f"sub_goal_{i}"]), i


@pytest.mark.gpu
def test_kernel_matches_oracle_and_reference():
    from benchnav_amd import NativeMPPI
    fx = _fx()
    p = _params(fx)
    with NativeMPPI(horizon=int(fx["T"]), num_samples=64, grid_size=int(fx["G"]), resolution=float(fx["res"]),
                    stuck_threshold=float(fx["thr"])) as pl:
        pl.set_map(fx["R"]); pl.set_goal(fx["goal"])
        for i in range(int(fx["n_solves"])):
            out = pl.dwa_solve(fx[f"state_{i}"], fx[f"actions_{i}"], fx[f"sub_goal_{i}"])
            orc = O.dwa(p, fx["R"], fx[f"state_{i}"], fx[f"actions_{i}"], fx[f"sub_goal_{i}"])
            assert np.array_equal(out["states"][0], orc["X"]) and np.array_equal(out["costs"][0], orc["cost"])
            assert int(out["best_index"][0]) == orc["best"] and np.abs(out["weights"][0] - orc["w"]).max() < 1e-6
            assert np.array_equal(out["best_action"][0], fx[f"actions_{i}"][orc["best"]])
            assert np.array_equal(out["best_states"][0], orc["X"][orc["best"]])
            assert np.abs(out["states"][0] - fx[f"X_{i}"]).max() <= 1e-4
            assert np.abs(out["weights"][0] - fx[f"w_{i}"]).max() <= 5e-3
        # no reference path: stage cost against the goal (dwa.py:243-247); ragged candidate count
        acts = fx["actions_0"][:37]
        o2 = pl.dwa_solve(fx["state_0"], acts)
        r2 = O.dwa(p, fx["R"], fx["state_0"], acts, None)
        assert np.array_equal(o2["costs"][0], r2["cost"]) and int(o2["best_index"][0]) == r2["best"]


@pytest.mark.gpu
def test_drop_in_class_follows_the_reference_run():
    """forward() is one asynchronous device call: window grid, sub-goal (from candidate 0's aliased slot), rollouts, costs,
    argmin.  Teacher-forced on the reference's previous action; every piece against the fixture."""
    import torch
    from helpers import FakeDynamics, FakeGridMap, FakeObjectives
    from benchnav_amd import DWA
    fx = _fx()
    gm = FakeGridMap(int(fx["G"]), float(fx["res"]))
    dyn = FakeDynamics(fx["R"], gm)
    obj = FakeObjectives(torch.tensor(fx["goal"]), float(fx["thr"]))
    solver = DWA(horizon=int(fx["T"]), dim_state=3, dim_control=2, dynamics=dyn, objectives=obj,
                 a_lim=torch.tensor(fx["a_lim"]), delta_t=float(fx["delta_t"]), lookahead_distance=float(fx["lookahead"]),
                 num_lin_vel=int(fx["nv"]), num_ang_vel=int(fx["nw"]))
    solver.update_reference_path(torch.tensor(fx["path"]))
    p = _params(fx)
    for i in range(int(fx["n_solves"])):
        state = torch.tensor(fx[f"state_{i}"])
        solver._previous_action_seq = torch.tensor(fx[f"prev_action_{i}"], device="cuda").unsqueeze(0)     # the REFERENCE's previous pick
        a_opt, x_opt = solver(state)
        assert a_opt.shape == (1, 2) and x_opt.shape == (1, int(fx["T"]) + 1, 3) and a_opt.is_cuda
        acts, sub_goal = solver.last_candidates()
        # torch.linspace itself differs in the last bit between AVX2 and AVX-512 hosts (vectorised head); the device uses the scalar form
        assert np.abs(acts.cpu().numpy() - fx[f"actions_{i}"]).max() <= 1e-6
        assert np.array_equal(sub_goal.cpu().numpy(), fx[f"sub_goal_{i}"]), i
        orc = O.dwa(p, fx["R"], fx[f"state_{i}"], acts.cpu().numpy(), sub_goal.cpu().numpy())
        assert np.array_equal(solver._state_seq_batch.cpu().numpy(), orc["X"]) and np.array_equal(solver._costs.cpu().numpy(), orc["cost"])
        assert np.array_equal(a_opt.cpu().numpy()[0], acts.cpu().numpy()[orc["best"]])
        assert np.abs(a_opt.cpu().numpy() - fx[f"a_opt_{i}"]).max() <= 1e-6
        assert np.abs(x_opt[0].cpu().numpy() - fx[f"x_opt_{i}"]).max() <= 1e-4
        assert torch.equal(solver._previous_action_seq, a_opt)                                # dwa.py:147
        top_s, top_w = solver.get_top_samples()
        assert top_s.shape == (100, int(fx["T"]) + 1, 3) and torch.all(top_w[:-1] >= top_w[1:])
    # free-running: the window follows the solver's own previous pick without touching the host
    solver._previous_action_seq = torch.zeros(int(fx["T"]), 2, device="cuda")
    state = torch.tensor(fx["state_0"], device="cuda")
    for i in range(3):
        a_opt, x_opt = solver(state)
        assert np.abs(a_opt.cpu().numpy() - fx[f"a_opt_{i}"]).max() <= 1e-6, i
        state = x_opt[0, 3].clone()
        state[2] = (state[2] + np.pi) % (2 * np.pi) - np.pi


@pytest.mark.gpu
def test_host_restatement_of_the_window_matches_the_device():
    import torch
    from helpers import FakeDynamics, FakeGridMap, FakeObjectives
    from benchnav_amd import DWA
    fx = _fx()
    dyn = FakeDynamics(fx["R"], FakeGridMap(int(fx["G"]), float(fx["res"])))
    solver = DWA(horizon=int(fx["T"]), dim_state=3, dim_control=2, dynamics=dyn, objectives=FakeObjectives(torch.tensor(fx["goal"]), float(fx["thr"])),
                 a_lim=torch.tensor(fx["a_lim"]), delta_t=float(fx["delta_t"]), num_lin_vel=7, num_ang_vel=13)
    rng = np.random.default_rng(0)
    for _ in range(20):
        prev = torch.tensor([[rng.uniform(0, 1), rng.uniform(-1, 1)]], dtype=torch.float32)
        solver._previous_action_seq = prev.cuda()
        solver(torch.tensor(fx["state_0"]))
        acts, sub_goal = solver.last_candidates()
        solver._previous_action_seq = prev.cuda()
        host = solver._generate_actions()
        assert host.shape == (91, 2) and np.abs(acts.cpu().numpy() - host.numpy()).max() <= 1e-6
        assert np.array_equal(sub_goal.cpu().numpy(), fx["goal"])           # no reference path: the goal
