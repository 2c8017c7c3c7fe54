"""PyTorch-CPU port of the reference MPPI solve step (TEST INFRASTRUCTURE ONLY).

Purpose: the `cpu_baseline` leg of bench.py ("kind": "port").  The reference is
Python/PyTorch and cannot travel to the GPU box, so this functional restatement
keeps the reference's execution structure -- one batch of ATen ops per time
step over the K rollouts, three T-long Python loops (rollout, costs, optimal
rollout) -- and is timed on the host cores beside the HIP path.  It is validated
against the golden fixtures in tests/test_torch_port.py.

Differences from the reference, none of which change results: no in-place
aliasing trick (slot t / slot t+1 are written explicitly), the
index origin tensor is built once instead of per lookup (grid_map.py:199-201).
Reference lines are cited per function.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch


@dataclass
class Problem:
    risk: torch.Tensor        # (G,G) [iy, ix]
    goal: torch.Tensor        # (2,)
    grid_size: int
    resolution: float
    x_limits: tuple
    y_limits: tuple
    sigmas: torch.Tensor      # (2,)
    lambda_: float
    stuck_threshold: float
    u_min: torch.Tensor
    u_max: torch.Tensor
    dt: float = 0.1

    def __post_init__(self):
        self.origin = torch.tensor([self.x_limits[0], self.y_limits[0]], dtype=torch.float32)
        self.inv_cov = torch.inverse(torch.diag(self.sigmas.to(torch.float32) ** 2))   # mppi.py:94-97
        self.goal = self.goal.to(torch.float32)


def traversability(pb: Problem, pos_xy: torch.Tensor) -> torch.Tensor:
    """grid_map.py:183-210 + :167 + traversability_model.py:70-72 for (N,2) positions."""
    idx = ((pos_xy - pb.origin) / pb.resolution).floor().int().clamp(0, pb.grid_size - 1)
    return 1 - torch.clamp(pb.risk[idx[:, 1], idx[:, 0]], 0, 1)


def transit(pb: Problem, state: torch.Tensor, action: torch.Tensor):
    """robot_model.py:59-100.  Returns (slot, next): the un-clamped/un-wrapped update the
    reference leaves in its input view, and the clamped/wrapped next state."""
    trav = traversability(pb, state[:, :2])
    x, y, theta = state.unbind(1)
    v, omega = action.unbind(1)
    v = torch.clamp(v, pb.u_min[0], pb.u_max[0])
    omega = torch.clamp(omega, pb.u_min[1], pb.u_max[1])
    xn = x + trav * v * torch.cos(theta) * pb.dt
    yn = y + trav * v * torch.sin(theta) * pb.dt
    tn = theta + trav * omega * pb.dt
    wrapped = (tn + torch.pi) % (2 * torch.pi) - torch.pi
    nxt = torch.stack([torch.clamp(xn, pb.x_limits[0], pb.x_limits[1]),
                       torch.clamp(yn, pb.y_limits[0], pb.y_limits[1]), wrapped], dim=1)
    return torch.stack([xn, yn, tn], dim=1), nxt


def stage_cost(pb: Problem, state: torch.Tensor) -> torch.Tensor:
    """objectives.py:29-53."""
    dist = torch.norm(state[:, :2] - pb.goal, dim=1)
    return dist + 1e4 * (traversability(pb, state[:, :2]) <= pb.stuck_threshold)


def rollout(pb: Problem, state0: torch.Tensor, actions: torch.Tensor) -> torch.Tensor:
    """mppi.py:160-165 / :202-214: (N,3) start, (N,T,2) controls -> (N,T+1,3) aliased slots."""
    N, T = actions.shape[0], actions.shape[1]
    seq = torch.zeros(N, T + 1, 3)
    cur = state0
    for t in range(T):
        slot, cur = transit(pb, cur, actions[:, t, :])
        seq[:, t, :] = slot
    seq[:, T, :] = cur
    return seq


@torch.no_grad()
def solve(pb: Problem, state: torch.Tensor, mean: torch.Tensor, eps: torch.Tensor | None = None, K: int | None = None):
    """mppi.py:130-219.  eps (K,T,2) standard normals, or None to draw them like rsample does."""
    T = mean.shape[0]
    if eps is None:
        eps = torch.empty(K, T, 2).normal_()                              # mppi.py:149-151
    K = eps.shape[0]
    U = torch.clamp(mean + eps * pb.sigmas, pb.u_min, pb.u_max)           # :152-157
    X = rollout(pb, state.to(torch.float32).repeat(K, 1), U)              # :160-165
    stage = torch.zeros(K, T)
    act = torch.zeros(K, T)
    for t in range(T):                                                    # :174-182
        stage[:, t] = stage_cost(pb, X[:, t, :])
        act[:, t] = mean[t] @ pb.inv_cov @ U[:, t].T
    term = stage_cost(pb, X[:, -1, :])                                    # :184
    cost = torch.sum(stage, dim=1) + term + torch.sum(pb.lambda_ * act, dim=1)
    w = torch.softmax(-cost / pb.lambda_, dim=0)                          # :193
    Ustar = torch.sum(w.view(K, 1, 1) * U, dim=0)                         # :196-199
    Xstar = rollout(pb, state.to(torch.float32).view(1, 3), Ustar.view(1, T, 2))   # :202-214
    return dict(U=U, X=X, cost=cost, w=w, Ustar=Ustar, Xstar=Xstar[0])


def problem_from_fixture(fx) -> Problem:
    return Problem(risk=torch.from_numpy(fx["R"]), goal=torch.from_numpy(fx["goal"]), grid_size=int(fx["G"]),
                   resolution=float(fx["res"]), x_limits=tuple(fx["x_limits"].tolist()),
                   y_limits=tuple(fx["y_limits"].tolist()), sigmas=torch.from_numpy(fx["sigmas"]),
                   lambda_=float(fx["lam"]), stuck_threshold=float(fx["thr"]),
                   u_min=torch.from_numpy(fx["u_min"]), u_max=torch.from_numpy(fx["u_max"]))
