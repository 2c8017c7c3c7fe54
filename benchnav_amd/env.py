"""B planetary environments on the device, with the reference environment's interface.

Mirror of `PlanetaryEnv` (src/simulator/planetary_env.py) for the part the planner loop touches -- `reset`,
`step`, `collision_check` (`:143-232`) -- batched over the B instances of a `NativeMPPI` handle: every return
value keeps the reference's shape with a leading batch dimension.  Rendering and the gym plumbing are out of
scope.  All tensors live on the planner's GPU; nothing returns to the host per control step.

    env = BatchedPlanetaryEnv(planner, latent_mean, latent_std, start_pos, goal_pos)
    state = env.reset(seed=0)                                   # (B, 3)
    while not done:
        planner.solve_async_device(state.data_ptr()); planner.flush()
        action = ustar_view[:, 0, :]                            # (B, 2) first control of every U*
        state, reward, is_terminated, is_truncated = env.step(action)
        is_collisions = env.collision_check(xstar_view)         # (B, T+1) bool

`env.run(n_steps)` is the fused form: the whole solve -> step loop inside the planner's pipelined launches.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np
import torch

from . import _capi


def env_inputs(env) -> dict:
    """What a BatchedPlanetaryEnv needs, read off a reference-shaped `PlanetaryEnv` (planetary_env.py:58-92): the latent slip
    model its observation-mode dynamics sample from (`grid_map.distributions["latent_models"]`, traversability_model.py:65-69),
    start / goal, time step and limit, the two thresholds, the seed, and the grid geometry.  Pinned against the real class by
    tests/golden/boundary.json (attribute names) and boundary.npz (values)."""
    gm = env._grid_map
    latent = gm.distributions["latent_models"]
    return dict(latent_mean=latent.mean, latent_std=latent.stddev, start_pos=env._start_pos, goal_pos=env._goal_pos,
                delta_t=float(env._delta_t), time_limit=float(env._time_limit), stuck_threshold=float(env.stuck_threshold),
                goal_threshold=float(env._goal_threshold), seed=env._seed, grid_size=int(gm.grid_size), resolution=float(gm.resolution),
                x_limits=(float(gm.x_limits[0]), float(gm.x_limits[1])), y_limits=(float(gm.y_limits[0]), float(gm.y_limits[1])))


class BatchedPlanetaryEnv:
    @classmethod
    def from_reference(cls, planner, env, freeze_on_goal: bool = False) -> "BatchedPlanetaryEnv":
        """B device-side copies of one reference `PlanetaryEnv` (same map, start, goal, thresholds), for a planner with B instances."""
        inp = env_inputs(env)
        if inp["grid_size"] != planner.G:
            raise ValueError(f"the environment's grid ({inp['grid_size']}) is not the planner's ({planner.G})")
        f32 = lambda t: t.detach().to("cpu", torch.float32).numpy()
        return cls(planner, f32(inp["latent_mean"]), f32(inp["latent_std"]), f32(inp["start_pos"]), f32(inp["goal_pos"]),
                   delta_t=inp["delta_t"], time_limit=inp["time_limit"], stuck_threshold=inp["stuck_threshold"],
                   goal_threshold=inp["goal_threshold"], seed=inp["seed"], freeze_on_goal=freeze_on_goal)

    def __init__(self, planner, latent_mean, latent_std, start_pos, goal_pos, delta_t: float = 0.1, time_limit: float = 100.0,
                 stuck_threshold: float = 0.1, goal_threshold: float = 1.0, seed: Optional[int] = None, freeze_on_goal: bool = False):
        """planner: NativeMPPI with B instances (its goals are set from goal_pos), created on torch's current stream of its
        device (`stream=torch.cuda.current_stream().cuda_stream`): the environment kernels run on the planner's stream and
        their inputs / outputs are torch tensors, so both must be ONE stream (checked on every call).
        latent_mean / latent_std: (G,G) slip model `grid_map.distributions["latent_models"]` (planetary_env.py:80-84 builds
        the observation-mode dynamics from it).  start_pos, goal_pos: (B,2) or (2,).
        freeze_on_goal: opt-in; the reference environment keeps moving when a terminated episode is stepped again."""
        self.planner = planner
        self._freeze = bool(freeze_on_goal)
        self._dev = torch.device("cuda", planner.device_id)
        self.B = planner.B
        self._lib = planner._lib
        self._h = planner._h
        self._delta_t, self._time_limit = float(delta_t), float(time_limit)
        self.stuck_threshold = float(stuck_threshold)
        self._goal_threshold = float(goal_threshold)
        self._seed = 0 if seed is None else int(seed)
        dev = self._dev
        self._check_stream()
        self._start_pos = torch.as_tensor(np.broadcast_to(np.asarray(start_pos, np.float32), (self.B, 2)).copy(), device=dev)
        self._goal_pos = torch.as_tensor(np.broadcast_to(np.asarray(goal_pos, np.float32), (self.B, 2)).copy(), device=dev)
        self._latent = (np.ascontiguousarray(latent_mean, np.float32), np.ascontiguousarray(latent_std, np.float32))
        for b in range(self.B):
            planner.set_goal(self._goal_pos[b].cpu().numpy(), b)
        planner.env_attach(self._latent[0], self._latent[1], goal_threshold=self._goal_threshold, delta_t=self._delta_t, seed=self._seed,
                           freeze_on_goal=self._freeze)
        self._robot_state = self._initialize_robot_state()
        self._reward = torch.full((self.B,), float("nan"), device=dev)
        self._terminated = torch.zeros(self.B, dtype=torch.int32, device=dev)
        self._elapsed_time = 0.0
        self._steps = 0
        self._draws = 0
        if bool(self.collision_check(torch.cat([self._start_pos, torch.zeros(self.B, 1, device=dev)], 1).unsqueeze(1)).any()) or \
           bool(self.collision_check(torch.cat([self._goal_pos, torch.zeros(self.B, 1, device=dev)], 1).unsqueeze(1)).any()):
            raise ValueError("Start or goal position is not traversable.")          # planetary_env.py:124-125

    def _check_stream(self):
        """The planner enqueues on the stream it was created with; torch produces / consumes the tensors on its current
        stream.  Without a shared stream nothing orders the two (a private planner stream would race the clone of the state,
        the action and the read-back), so a mismatch is an error, not a silent hazard."""
        cur = torch.cuda.current_stream(self._dev).cuda_stream
        if self.planner.stream is None or int(self.planner.stream) != int(cur):
            raise RuntimeError("BatchedPlanetaryEnv needs a planner created on torch's current stream of its device "
                               "(NativeMPPI(..., stream=torch.cuda.current_stream().cuda_stream)); got planner stream "
                               f"{self.planner.stream!r}, current stream {cur}")

    def _initialize_robot_state(self) -> torch.Tensor:
        """(x, y, theta) with the heading towards the goal (planetary_env.py:128-141)."""
        d = self._goal_pos - self._start_pos
        return torch.cat([self._start_pos, torch.atan2(d[:, 1], d[:, 0]).unsqueeze(1)], 1).contiguous()

    def reset(self, seed: Optional[int] = None) -> torch.Tensor:
        """planetary_env.py:143-187: elapsed time, robot state and reward back to their initial values; returns (B,3)."""
        if seed is not None:
            self._seed = int(seed)
            self.planner.env_attach(self._latent[0], self._latent[1], goal_threshold=self._goal_threshold, delta_t=self._delta_t,
                                    seed=self._seed, freeze_on_goal=self._freeze)
        self._elapsed_time, self._steps, self._draws = 0.0, 0, 0
        self._robot_state = self._initialize_robot_state()
        self._reward.fill_(float("nan"))
        self._terminated.zero_()
        return self._robot_state

    def step(self, action: torch.Tensor, z: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, bool]:
        """planetary_env.py:189-219 for every environment.  action (B,2) on the GPU.  Returns (robot_state (B,3),
        reward (B,) = sampled traversability, is_terminated (B,) bool, is_truncated bool).  z (B,) injects the slip draws."""
        self._check_stream()
        a = action.to(self._dev, torch.float32).contiguous()
        assert a.shape == (self.B, 2)
        zp = None if z is None else z.to(self._dev, torch.float32).contiguous()
        state = self._robot_state.clone()                          # the reference returns a new tensor every step
        _capi.check(self._lib.bn_mppi_env_step(self._h, C.c_void_p(a.data_ptr()), C.c_void_p(state.data_ptr()),
                                               C.c_void_p(self._reward.data_ptr()), C.c_void_p(self._terminated.data_ptr()),
                                               C.c_void_p(None if zp is None else zp.data_ptr()), self._steps))
        self._keep = (a, zp)
        self._robot_state = state
        self._steps += 1
        self._elapsed_time += self._delta_t
        return self._robot_state, self._reward, self._terminated.bool(), self._elapsed_time > self._time_limit

    def collision_check(self, states: torch.Tensor, z: Optional[torch.Tensor] = None) -> torch.Tensor:
        """planetary_env.py:221-232: states (B, N, 3) -> is_collisions (B, N) bool, one fresh slip draw per position."""
        self._check_stream()
        s = states.to(self._dev, torch.float32).contiguous()
        assert s.dim() == 3 and s.shape[0] == self.B and s.shape[2] == 3
        out = torch.empty(self.B, s.shape[1], dtype=torch.uint8, device=self._dev)
        zp = None if z is None else z.to(self._dev, torch.float32).contiguous()
        _capi.check(self._lib.bn_mppi_env_collision_check(self._h, C.c_void_p(s.data_ptr()), s.shape[1], self.stuck_threshold,
                                                          C.c_void_p(None if zp is None else zp.data_ptr()), self._draws,
                                                          C.c_void_p(out.data_ptr())))
        self._keep_c = (s, zp)
        self._draws += 1
        return out.bool()

    def run(self, n_steps: int):
        """The fused loop: n_steps of solve -> step inside the planner's pipelined launches (bn_mppi_episode_async),
        from the current robot states.  Returns (states (n_steps+1,B,3), rewards (n_steps,B), first_goal_step (B,))."""
        states, rewards, done = self.planner.episode(n_steps, self._robot_state.cpu().numpy())
        self._robot_state = torch.as_tensor(states[-1], device=self._dev).contiguous()
        self._steps += n_steps
        self._elapsed_time += n_steps * self._delta_t
        return states, rewards, done
