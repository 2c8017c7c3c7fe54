"""MI355X-native Dynamic Window Approach with the reference's Python interface ("next" row N3).

Mirror of `DWA(nn.Module)` in the reference's src/planners/local_planners/dwa.py (constructor :17-31,
forward :116, update_reference_path :155, get_top_samples :287).  The candidate rollouts, costs, argmin and
weights run in the HIP kernel behind bn_mppi_dwa_solve; the two tiny pieces of host geometry -- the dynamic
window grid (dwa.py:168-199) and the sub-goal pick on the reference path (dwa.py:260-285) -- are computed with
torch on the CPU as the reference does, so the candidate set is the reference's, value for value.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from .mppi import _DevArray, _planner_inputs
from .native import NativeMPPI


class DWA(nn.Module):
    def __init__(self, horizon: int, dim_state: int, dim_control: int, dynamics, objectives, a_lim: torch.Tensor,
                 delta_t: float, lookahead_distance: float = 1.0, num_lin_vel: int = 10, num_ang_vel: int = 10,
                 device=torch.device("cuda"), dtype=torch.float32, seed: int = 42) -> None:
        super().__init__()
        torch.manual_seed(seed)                                        # dwa.py:55
        assert dynamics.min_action.shape == (dim_control,), "minimum actions must be a tensor of shape (dim_control,)"
        assert dynamics.max_action.shape == (dim_control,), "maximum actions must be a tensor of shape (dim_control,)"
        assert a_lim.shape == (dim_control,), "acceleration limits must be a tensor of shape (dim_control,)"
        if dim_state != 3 or dim_control != 2 or dtype != torch.float32:
            raise ValueError("the native planner implements the float32 unicycle model: dim_state=3, dim_control=2")
        if not torch.cuda.is_available():
            raise RuntimeError("benchnav_amd.DWA needs an MI355X (gfx950) device; there is no CPU fallback")
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError(f"benchnav_amd.DWA runs on the GPU only (device={device!r})")
        self._device = torch.device("cuda", torch.cuda.current_device()) if dev.index is None else dev
        self._dtype = dtype
        self._horizon, self._dim_state, self._dim_control = horizon, dim_state, dim_control
        self._dynamics = dynamics
        self._u_min = dynamics.min_action.detach().to("cpu", dtype).clone()
        self._u_max = dynamics.max_action.detach().to("cpu", dtype).clone()
        self._a_lim = a_lim.detach().to("cpu", dtype).clone()
        self._delta_t = delta_t
        self._lookahead_distance = lookahead_distance
        self._num_lin_vel, self._num_ang_vel = num_lin_vel, num_ang_vel
        inp = _planner_inputs(dynamics, objectives)
        self._goal = torch.as_tensor(inp["goal"]).detach().to("cpu", dtype)
        self._native = NativeMPPI(horizon=horizon, num_samples=64, grid_size=inp["grid_size"], resolution=inp["resolution"],
                                  x_limits=inp["x_limits"], y_limits=inp["y_limits"],
                                  u_min=self._u_min.tolist(), u_max=self._u_max.tolist(),
                                  stuck_threshold=inp["stuck_threshold"], device_id=self._device.index, stream=0)
        self._risk_cpu = inp["risks"].detach().to("cpu", torch.float32).contiguous()
        self._grid = (inp["grid_size"], inp["resolution"], inp["x_limits"], inp["y_limits"])
        self._native.set_map(self._risk_cpu.numpy())
        self._native.set_goal(self._goal.numpy())
        n = num_lin_vel * num_ang_vel
        self._previous_action_seq = torch.zeros(horizon, dim_control, device=self._device, dtype=dtype)
        self._state_seq_batch = torch.zeros(n, horizon + 1, dim_state, device=self._device, dtype=dtype)
        self._weights = torch.zeros(n, device=self._device, dtype=dtype)
        self.reference_path: Optional[torch.Tensor] = None

    # -- host geometry (tiny; torch-CPU like the reference so the numbers are the reference's) -----------------
    def _generate_actions(self) -> torch.Tensor:
        """The dynamic window around the previous first control, as an (nv*nw, 2) grid (dwa.py:168-199)."""
        prev = self._previous_action_seq[0, :].detach().to("cpu", self._dtype)
        reach = self._a_lim * self._delta_t
        lo = torch.maximum(self._u_min, prev - reach)
        hi = torch.minimum(self._u_max, prev + reach)
        vs = torch.linspace(lo[0], hi[0], self._num_lin_vel, dtype=self._dtype)
        ws = torch.linspace(lo[1], hi[1], self._num_ang_vel, dtype=self._dtype)
        return torch.cartesian_prod(vs, ws)

    def _select_sub_goal(self, state: torch.Tensor) -> torch.Tensor:
        """Nearest reference-path point that lies ahead (|bearing| < 90 deg) and beyond the look-ahead distance,
        else the path's end (dwa.py:260-285)."""
        path = self.reference_path
        d = path - state[:2]
        dist = torch.norm(d, dim=1)
        bearing = torch.atan2(d[:, 1], d[:, 0]) - state[2]
        ahead = (bearing.abs() < torch.pi / 2) & (dist > self._lookahead_distance)
        if ahead.any():
            nearest = dist[ahead].min()
            return path[torch.where(dist == nearest)[0][0]]
        return path[-1]

    def _sub_goal_state(self, state: torch.Tensor, action0: torch.Tensor) -> torch.Tensor:
        """The state the reference's sub-goal rule sees.  _compute_costs calls _select_sub_goal(state_seq_batch[0, 0, :])
        AFTER the rollouts (dwa.py:240-244), and transit's in-place `x +=` / `theta +=` (robot_model.py:86-88) has by then
        turned candidate 0's slot 0 into the input state advanced by one un-clamped, un-wrapped step of candidate 0.
        Same torch-CPU operations in the same order as robot_model.py:75-88 / grid_map.py:195-209 (transit's default
        delta_t = 0.1: DWA does not pass its own)."""
        G, res, xl, yl = self._grid
        origin = torch.tensor([xl[0], yl[0]], dtype=self._dtype)
        idx = ((state[:2] - origin) / res).floor().int().clamp(0, G - 1)
        trav = 1 - torch.clamp(self._risk_cpu[idx[1], idx[0]], 0, 1)
        v = torch.clamp(action0[0], self._u_min[0], self._u_max[0])
        omega = torch.clamp(action0[1], self._u_min[1], self._u_max[1])
        x = state[0] + trav * v * torch.cos(state[2]) * 0.1
        y = state[1] + trav * v * torch.sin(state[2]) * 0.1
        theta = state[2] + trav * omega * 0.1
        return torch.stack([x, y, theta])

    def update_reference_path(self, reference_path: torch.Tensor) -> None:
        if reference_path is not None:
            assert reference_path.shape[1] == 2, "reference_path must be a tensor of shape (num_positions, 2)"
            self.reference_path = reference_path.detach().to("cpu", self._dtype)

    # -- the solve -------------------------------------------------------------------------------------------------
    def forward(self, state: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Returns (optimal_action_seq (1,2), optimal_state_seq (1,T+1,3)) on the planner's device (dwa.py:116-153)."""
        if not torch.is_tensor(state):
            state = torch.tensor(state, dtype=self._dtype)
        assert state.shape == (self._dim_state,), "state must be a tensor of shape (dim_state,)"
        st = state.detach().to("cpu", self._dtype)
        actions = self._generate_actions()
        sub_goal = self._select_sub_goal(self._sub_goal_state(st, actions[0])) if self.reference_path is not None else None
        out = self._native.dwa_solve(st.numpy(), actions.numpy(), None if sub_goal is None else sub_goal.numpy(), full=False)
        best = int(out["best_index"][0])
        n = actions.shape[0]
        optimal_action_seq = actions[best].unsqueeze(0).to(self._device)
        optimal_state_seq = torch.from_numpy(out["best_states"]).to(self._device)
        self._previous_action_seq = optimal_action_seq                     # dwa.py:147
        # the candidate batch stays on the device: views of the library's buffers, copied into tensors the caller may keep
        xp, cp, wp = self._native.dwa_buffers(n)
        self._state_seq_batch = torch.as_tensor(_DevArray(xp, (n, self._horizon + 1, 3)), device=self._device).clone()
        self._costs = torch.as_tensor(_DevArray(cp, (n,)), device=self._device).clone()
        self._weights = torch.as_tensor(_DevArray(wp, (n,)), device=self._device).clone()
        return optimal_action_seq, optimal_state_seq

    def get_top_samples(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """All candidates sorted by weight, best first (dwa.py:287-299)."""
        order = torch.argsort(self._weights, descending=True)
        return self._state_seq_batch[order], self._weights[order]
