"""MI355X-native Dynamic Window Approach with the reference's Python interface ("next" row N3).

Mirror of `DWA(nn.Module)` in the reference's src/planners/local_planners/dwa.py (constructor :17-31,
forward :116, update_reference_path :155, get_top_samples :287).  forward() is ONE asynchronous C call
(bn_mppi_dwa_forward_async): the dynamic window grid (dwa.py:168-199), the sub-goal pick on the reference path
(dwa.py:240-244, 260-285: from candidate 0's aliased slot-0 state, as the reference does), the candidate rollouts,
costs, argmin and weights all run on the device; nothing returns to the host.  `_generate_actions`, `_sub_goal_state`
and `_select_sub_goal` restate the two pieces of geometry with torch-CPU operations in the reference's order (what the
tests compare the device path with); forward() does not call them.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

import ctypes as C

from . import _capi
from .mppi import MPPI as _MPPI, _DevArray, _planner_inputs
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
        self._stream = torch.cuda.current_stream(self._device)         # the library enqueues here; other current streams are fenced
        self._native = NativeMPPI(horizon=horizon, num_samples=64, grid_size=inp["grid_size"], resolution=inp["resolution"],
                                  x_limits=inp["x_limits"], y_limits=inp["y_limits"],
                                  u_min=self._u_min.tolist(), u_max=self._u_max.tolist(),
                                  stuck_threshold=inp["stuck_threshold"], device_id=self._device.index,
                                  stream=self._stream.cuda_stream)
        self._risk_cpu = inp["risks"].detach().to("cpu", torch.float32).contiguous()
        self._grid = (inp["grid_size"], inp["resolution"], inp["x_limits"], inp["y_limits"])
        self._native.set_map(self._risk_cpu.numpy())
        self._native.set_goal(self._goal.numpy())
        self._previous_action_seq = torch.zeros(horizon, dim_control, device=self._device, dtype=dtype)
        self._prev_buf = torch.zeros(1, dim_control, device=self._device, dtype=dtype)      # the window's centre, updated by the kernel
        self._prev_seen = (None, -1)                # (tensor object, its version) of the _previous_action_seq the buffer already mirrors
        self._a_lim_c = (C.c_float * 2)(float(self._a_lim[0]), float(self._a_lim[1]))
        self._solved = False
        self.reference_path: Optional[torch.Tensor] = None
        self._path_dev: Optional[torch.Tensor] = None

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
            self._path_dev = self.reference_path.to(self._device).contiguous()

    # -- the solve -------------------------------------------------------------------------------------------------
    def forward(self, state: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Returns (optimal_action_seq (1,2), optimal_state_seq (1,T+1,3)) on the planner's device (dwa.py:116-153),
        stream-ordered like any torch op: one C call, no host synchronisation."""
        if not torch.is_tensor(state):
            state = torch.tensor(state, dtype=self._dtype)
        assert state.shape == (self._dim_state,), "state must be a tensor of shape (dim_state,)"
        if state.device != self._device or state.dtype != self._dtype or not state.is_contiguous():
            state = state.detach().to(self._device, self._dtype).contiguous()
        prev = self._previous_action_seq
        x_opt = torch.empty(1, self._horizon + 1, 3, device=self._device, dtype=self._dtype)
        path = self._path_dev
        # Called under another current stream than the one captured at construction: fence the two (as MPPI.forward does), so
        # that the state / window-centre reads and the clone below are ordered with the kernels.
        with _MPPI._Fence(self):
            with torch.cuda.stream(self._stream):
                if prev is not self._prev_seen[0] or prev._version != self._prev_seen[1]:      # assigned or modified by the caller (or the
                    self._prev_buf.copy_(prev[:1].to(self._device, self._dtype))               # initial zeros): mirror its first row
                _capi.check(self._native._lib.bn_mppi_dwa_forward_async(
                    self._native._h, state.data_ptr(), self._prev_buf.data_ptr(), self._a_lim_c, self._delta_t, self._num_lin_vel,
                    self._num_ang_vel, None if path is None else path.data_ptr(), 0 if path is None else path.shape[0],
                    self._lookahead_distance, x_opt.data_ptr()))
                optimal_action_seq = self._prev_buf.clone()             # (1,2): the argmin action, written by the kernel
        self._keep = state
        self._previous_action_seq = optimal_action_seq                  # dwa.py:147
        self._prev_seen = (optimal_action_seq, optimal_action_seq._version)
        self._solved = True
        return optimal_action_seq, x_opt

    def _scratch_views(self):
        """Views of the library's scratch block: valid until the next forward() (which overwrites, and may regrow, it).
        Internal; the public attributes below hand out copies like the reference's fresh tensors (dwa.py:148-149)."""
        n = self._num_lin_vel * self._num_ang_vel
        xp, cp, wp = self._native.dwa_buffers(n)
        return (torch.as_tensor(_DevArray(xp, (n, self._horizon + 1, 3)), device=self._device),
                torch.as_tensor(_DevArray(cp, (n,)), device=self._device), torch.as_tensor(_DevArray(wp, (n,)), device=self._device))

    @property
    def _state_seq_batch(self) -> torch.Tensor:
        """(n, T+1, 3) candidate trajectories of the latest forward() (a copy: the library's buffer is reused by the next call)."""
        if not self._solved:
            return torch.zeros(self._num_lin_vel * self._num_ang_vel, self._horizon + 1, self._dim_state, device=self._device, dtype=self._dtype)
        with _MPPI._Fence(self), torch.cuda.stream(self._stream):
            return self._scratch_views()[0].clone()

    @property
    def _costs(self) -> torch.Tensor:
        with _MPPI._Fence(self), torch.cuda.stream(self._stream):
            return self._scratch_views()[1].clone()

    @property
    def _weights(self) -> torch.Tensor:
        if not self._solved:
            return torch.zeros(self._num_lin_vel * self._num_ang_vel, device=self._device, dtype=self._dtype)
        with _MPPI._Fence(self), torch.cuda.stream(self._stream):
            return self._scratch_views()[2].clone()

    def last_candidates(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(actions (n,2), sub_goal (2,)) the latest forward() used, as computed on the device."""
        n = self._num_lin_vel * self._num_ang_vel
        a, g = C.c_void_p(), C.c_void_p()
        _capi.check(self._native._lib.bn_mppi_dwa_candidates(self._native._h, n, C.byref(a), C.byref(g)))
        with _MPPI._Fence(self), torch.cuda.stream(self._stream):
            return (torch.as_tensor(_DevArray(a.value, (n, 2)), device=self._device).clone(),
                    torch.as_tensor(_DevArray(g.value, (2,)), device=self._device).clone())

    def get_top_samples(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """All candidates sorted by weight, best first (dwa.py:287-299)."""
        with _MPPI._Fence(self), torch.cuda.stream(self._stream):
            X, _, w = self._scratch_views()
            order = torch.argsort(w, descending=True)
            return X[order], w[order]
