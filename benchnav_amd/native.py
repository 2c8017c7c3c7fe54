"""NumPy-level wrapper of the C ABI (no torch): one handle, B instances.

This is the thinnest host layer above include/benchnav_mppi.h; the torch-facing
`benchnav_amd.MPPI` mirrors the reference class, this one serves C-ABI level
tests, the benchmark and batched (multi-instance) solves.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _capi


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f32(a, shape=None) -> np.ndarray:
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    if shape is not None and a.shape != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {a.shape}")
    return a


class NativeMPPI:
    def __init__(self, *, horizon: int, num_samples: int, grid_size: int, resolution: float,
                 x_limits: Optional[Sequence[float]] = None, y_limits: Optional[Sequence[float]] = None,
                 sigmas=(0.5, 0.5), inv_var=None, lambda_: float = 0.5, u_min=(0.0, -1.0), u_max=(1.0, 1.0),
                 dt: float = 0.1, stuck_threshold: float = 0.3, num_instances: int = 1, shared_map: bool = False,
                 seed: int = 42, device_id: int = 0, store_controls: bool = False, lds_window: bool = True,
                 profile: bool = False, stream: Optional[int] = None, pipeline: bool = True, sampled_slip: bool = False, kernel: str = "auto",
                 lean: bool = False, overlap: bool = True, reference_order: bool = False, host_paced: bool = False, unordered_outputs: bool = False):
        self._lib = _capi.load()
        self._h = C.c_void_p()
        cfg = _capi.Config()
        self._lib.bn_mppi_config_init(C.byref(cfg))
        if x_limits is None:       # reference GridMap geometry, grid_map.py:42-50
            c = grid_size * resolution / 2
            x_limits = (c - grid_size / 2 * resolution, c + grid_size / 2 * resolution)
        if y_limits is None:
            y_limits = x_limits
        if inv_var is None:
            s32 = np.asarray(sigmas, np.float32)
            inv_var = (np.float32(1) / (s32 * s32)).tolist()
        cfg.device_id = device_id
        cfg.horizon, cfg.num_samples, cfg.num_instances = horizon, num_samples, num_instances
        cfg.grid_size, cfg.resolution = grid_size, resolution
        for i in range(2):
            cfg.x_limits[i], cfg.y_limits[i] = x_limits[i], y_limits[i]
            cfg.sigma[i], cfg.inv_var[i] = sigmas[i], inv_var[i]
            cfg.u_min[i], cfg.u_max[i] = u_min[i], u_max[i]
        cfg.lambda_, cfg.dt, cfg.stuck_threshold, cfg.seed = lambda_, dt, stuck_threshold, seed
        cfg.flags = ((_capi.BN_FLAG_STORE_CONTROLS if store_controls else 0)
                     | (_capi.BN_FLAG_SHARED_MAP if shared_map else 0)
                     | (0 if lds_window else _capi.BN_FLAG_NO_LDS_WINDOW)
                     | (_capi.BN_FLAG_PROFILE if profile else 0)
                     | (_capi.BN_FLAG_PRIVATE_STREAM if stream is None else 0)
                     | (0 if pipeline else _capi.BN_FLAG_NO_PIPELINE)
                     | (_capi.BN_FLAG_SAMPLED_SLIP if sampled_slip else 0)
                     | (_capi.BN_FLAG_LEAN if lean else 0)
                     | (0 if overlap else _capi.BN_FLAG_NO_OVERLAP)
                     | (_capi.BN_FLAG_REFERENCE_ORDER if reference_order else 0)
                     | (_capi.BN_FLAG_HOST_PACED if host_paced else 0)
                     | (_capi.BN_FLAG_UNORDERED_OUTPUTS if unordered_outputs else 0)
                     | {"auto": 0, "wave": _capi.BN_FLAG_WAVE_KERNEL, "role": _capi.BN_FLAG_ROLE_KERNEL, "lat": _capi.BN_FLAG_LAT_KERNEL}[kernel])
        cfg.stream = stream        # an int hipStream_t; 0 is the null stream (torch's default); None = private stream
        self.K, self.T, self.G, self.B = num_samples, horizon, grid_size, num_instances
        self.device_id, self.stream = device_id, stream       # stream: the hipStream_t the handle enqueues on (None: private)
        self.store_controls = store_controls
        self._shared_map = shared_map or num_instances == 1
        self._ep_steps = 0
        _capi.check(self._lib.bn_mppi_create(C.byref(cfg), C.byref(self._h)))

    def close(self):
        if self._h is not None and self._h.value:
            self._lib.bn_mppi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- inputs ----------------------------------------------------------------------
    def set_map(self, risk, instance: int = -1):
        r = _f32(risk, (self.G, self.G))
        _capi.check(self._lib.bn_mppi_set_map(self._h, instance, C.c_void_p(r.ctypes.data), _capi.BN_MEM_HOST))

    def set_slip_std(self, std, instance: int = -1):
        """Sampled-slip mode: per-cell slip std; the map given to set_map is then the slip mean."""
        r = _f32(std, (self.G, self.G))
        _capi.check(self._lib.bn_mppi_set_slip_std(self._h, instance, C.c_void_p(r.ctypes.data), _capi.BN_MEM_HOST))

    def set_slip_noise(self, zt_ptr: Optional[int], zc_ptr: Optional[int], zo_ptr: Optional[int]):
        """Injected standard normals (device pointers): transit (B,T,K), cost (B,T+1,K), optimal rollout (B,T); None = Philox."""
        _capi.check(self._lib.bn_mppi_set_slip_noise(self._h, C.c_void_p(zt_ptr), C.c_void_p(zc_ptr), C.c_void_p(zo_ptr)))

    def slip_noise(self, solve_index: int, instance: int = 0):
        """The library's Philox slip draws of a solve: zt (K,T), zc (K,T+1), zo (T)."""
        zt = np.empty((self.K, self.T), np.float32); zc = np.empty((self.K, self.T + 1), np.float32); zo = np.empty(self.T, np.float32)
        _capi.check(self._lib.bn_mppi_get_slip_noise(self._h, instance, solve_index, _fp(zt), _fp(zc), _fp(zo)))
        return zt, zc, zo

    def set_goal(self, goal, instance: int = -1):
        _capi.check(self._lib.bn_mppi_set_goal(self._h, instance, _fp(_f32(goal, (2,)))))

    def set_mean(self, mean=None, instance: int = -1):
        if mean is None:
            _capi.check(self._lib.bn_mppi_set_mean(self._h, instance, None))
        else:
            _capi.check(self._lib.bn_mppi_set_mean(self._h, instance, _fp(_f32(mean, (self.T, 2)))))

    def first_action(self, instance: int = 0) -> np.ndarray:
        """U*[0] of the latest solve on the host, as soon as the tail has merged it (no stream synchronisation, no copy)."""
        out = np.empty(2, np.float32)
        _capi.check(self._lib.bn_mppi_first_action(self._h, instance, _fp(out)))
        return out

    def get_mean(self, instance: int = 0) -> np.ndarray:
        out = np.empty((self.T, 2), np.float32)
        _capi.check(self._lib.bn_mppi_get_mean(self._h, instance, _fp(out)))
        return out

    # -- solve -------------------------------------------------------------------------
    def solve(self, states, eps=None):
        """Synchronous solve of all B instances.  eps: None (Philox) or (B,K,T,2) / (K,T,2) host noise.
        Returns (Ustar (B,T,2), Xstar (B,T+1,3))."""
        st = _f32(states).reshape(self.B, 3)
        us = np.empty((self.B, self.T, 2), np.float32)
        xs = np.empty((self.B, self.T + 1, 3), np.float32)
        if eps is None:
            kind, eptr = _capi.BN_NOISE_PHILOX, C.c_void_p(None)
        else:
            e = _f32(eps).reshape(self.B, self.K, self.T, 2)
            kind, eptr = _capi.BN_NOISE_HOST_KT2, C.c_void_p(e.ctypes.data)
        _capi.check(self._lib.bn_mppi_solve(self._h, C.c_void_p(st.ctypes.data), _capi.BN_MEM_HOST, eptr, kind,
                                            _fp(us), _fp(xs)))
        return us, xs

    def solve_async_device(self, state_ptr: int, eps_ptr: Optional[int] = None, kind: int = _capi.BN_NOISE_PHILOX):
        """Enqueue one solve with device-resident state (and noise); nothing is copied."""
        _capi.check(self._lib.bn_mppi_solve_async(self._h, C.c_void_p(state_ptr), _capi.BN_MEM_DEVICE,
                                                  C.c_void_p(eps_ptr), kind))

    def forward_async_device(self, state_ptr: int, eps_ptr: Optional[int] = None, kind: int = _capi.BN_NOISE_PHILOX, out_ptr: Optional[int] = None):
        """MPPI.forward as one call: solve + tail enqueued (ONE launch on the latency kernel, launches_per_forward())."""
        _capi.check(self._lib.bn_mppi_forward_async(self._h, C.c_void_p(state_ptr), C.c_void_p(eps_ptr), kind, C.c_void_p(out_ptr)))

    def forward_state_async(self, states, eps_ptr: Optional[int] = None, kind: int = _capi.BN_NOISE_PHILOX, out_ptr: Optional[int] = None):
        """... with the (B,3) states taken from the host by value at the call (no upload is enqueued for one instance on the latency kernel)."""
        st = _f32(states).reshape(self.B, 3)
        _capi.check(self._lib.bn_mppi_forward_state_async(self._h, C.c_void_p(st.ctypes.data), C.c_void_p(eps_ptr), kind, C.c_void_p(out_ptr)))

    def host_paced(self) -> bool:
        """BN_FLAG_HOST_PACED was given and the handle qualifies: forward_state_async enqueues the next solve's launch one step ahead."""
        return int(self._lib.bn_mppi_host_paced(self._h)) >= 1

    def order_outputs(self) -> None:
        """BN_FLAG_UNORDERED_OUTPUTS: order the handle's stream behind the latest posted solve (before the caller's own consumers)."""
        _capi.check(self._lib.bn_mppi_order_outputs(self._h))

    def states_buffer_index(self) -> int:
        return int(self._lib.bn_mppi_states_buffer_index(self._h))

    def launches_per_forward(self) -> int:
        return int(self._lib.bn_mppi_launches_per_forward(self._h))

    def solve_n_async_device(self, n: int, state_ptr: int, eps_ptr: Optional[int] = None,
                             kind: int = _capi.BN_NOISE_PHILOX, eps_ring: int = 1, eps_stride: int = 0):
        """Enqueue n dependent (warm-started) solves from one C call: no per-launch Python overhead."""
        _capi.check(self._lib.bn_mppi_solve_n_async(self._h, n, C.c_void_p(state_ptr), _capi.BN_MEM_DEVICE,
                                                    C.c_void_p(eps_ptr), kind, eps_ring, eps_stride))

    # -- DWA on the same transit / cost kernels -------------------------------------------------
    # ---- K-sharded solve (one solve split over ranks; see benchnav_amd.sharding.ShardedMPPI) ----
    def set_rollout_offset(self, first_rollout: int):
        _capi.check(self._lib.bn_mppi_set_rollout_offset(self._h, first_rollout))

    def shard_rollout_async_device(self, state_ptr: int, eps_ptr: Optional[int] = None, kind: int = _capi.BN_NOISE_PHILOX):
        _capi.check(self._lib.bn_mppi_shard_rollout_async(self._h, C.c_void_p(state_ptr), _capi.BN_MEM_DEVICE,
                                                          C.c_void_p(eps_ptr), kind))

    def shard_comm_prepare(self, world_size: int, rank: int):
        """The local half of shard_comm_init (RCCL opened, buffers, events, side stream): no communication, may fail on one rank only."""
        _capi.check(self._lib.bn_mppi_shard_comm_prepare(self._h, world_size, rank))

    def shard_comm_init(self, unique_id: bytes, world_size: int, rank: int):
        """Collective: the ranks of a K-sharded solve build the RCCL communicator the library enqueues its exchange on."""
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        _capi.check(self._lib.bn_mppi_shard_comm_init(self._h, C.cast(buf, C.c_void_p), world_size, rank))

    def shard_solve_async_device(self, state_ptr: int, eps_ptr: Optional[int] = None, kind: int = _capi.BN_NOISE_PHILOX):
        """Rollouts of the shard, all-gather of the partial rows (RCCL on the handle's stream), merge + tail: one call, one queue."""
        _capi.check(self._lib.bn_mppi_shard_solve_async(self._h, C.c_void_p(state_ptr), _capi.BN_MEM_DEVICE, C.c_void_p(eps_ptr), kind))

    def shard_partials(self):
        """(device pointer, workgroups, floats per workgroup) of the shard's softmin partials."""
        ptr, n, ps = C.c_void_p(), C.c_int32(), C.c_int32()
        _capi.check(self._lib.bn_mppi_shard_partials(self._h, C.byref(ptr), C.byref(n), C.byref(ps)))
        return ptr.value, n.value, ps.value

    def shard_finish_async(self, all_partials_ptr: int, total_workgroups: int):
        _capi.check(self._lib.bn_mppi_shard_finish_async(self._h, C.c_void_p(all_partials_ptr), total_workgroups))

    def dwa_solve(self, states, actions, stage_goal=None, full: bool = True):
        """Roll out and cost constant-control candidates `actions` (B,NA,2) or (NA,2); see bn_mppi_dwa_solve.
        Returns dict(best_action (B,2), best_states (B,T+1,3), costs, weights (B,NA), states (B,NA,T+1,3), best_index (B));
        full=False leaves costs / weights / states on the device (dwa_buffers) and returns the small outputs only."""
        st = _f32(states).reshape(self.B, 3)
        act = _f32(actions)
        if act.ndim == 2:
            act = np.ascontiguousarray(np.broadcast_to(act, (self.B,) + act.shape))
        NA = act.shape[1]
        assert act.shape == (self.B, NA, 2)
        sg = None if stage_goal is None else _f32(stage_goal).reshape(self.B, 2)
        out = dict(best_action=np.empty((self.B, 2), np.float32), best_states=np.empty((self.B, self.T + 1, 3), np.float32),
                   costs=np.empty((self.B, NA), np.float32), weights=np.empty((self.B, NA), np.float32),
                   states=np.empty((self.B, NA, self.T + 1, 3), np.float32), best_index=np.empty(self.B, np.int32))
        _capi.check(self._lib.bn_mppi_dwa_solve(self._h, _fp(st), _fp(act), NA, None if sg is None else _fp(sg),
                                                _fp(out["best_action"]), _fp(out["best_states"]), _fp(out["costs"]) if full else None,
                                                _fp(out["weights"]) if full else None, _fp(out["states"]) if full else None,
                                                out["best_index"].ctypes.data_as(C.POINTER(C.c_int32))))
        if not full:
            for k in ("costs", "weights", "states"):
                del out[k]
        return out

    def dwa_buffers(self, num_actions: int):
        """Device pointers (states_all, costs, weights) of the latest dwa_solve."""
        x, c, w = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _capi.check(self._lib.bn_mppi_dwa_buffers(self._h, num_actions, C.byref(x), C.byref(c), C.byref(w)))
        return x.value, c.value, w.value

    # -- device-side closed loop (PlanetaryEnv.step between solves) ------------------------------
    def env_attach(self, latent_mean, latent_std, goal_threshold: float = 1.0, delta_t: float = 0.1, seed: int = 0,
                   freeze_on_goal: bool = False):
        """Latent slip model Normal(mean, std) per cell ((G,G) shared or (n_maps,G,G)), PlanetaryEnv defaults
        (planetary_env.py:36: goal_threshold=1.0).  freeze_on_goal: an instance within goal_threshold stays put (opt-in;
        the reference environment keeps moving when a terminated episode is stepped)."""
        n_maps = 1 if self._shared_map else self.B
        m = _f32(latent_mean).reshape(-1, self.G, self.G)
        s = _f32(latent_std).reshape(-1, self.G, self.G)
        if m.shape[0] == 1 and n_maps > 1:
            m = np.ascontiguousarray(np.repeat(m, n_maps, axis=0)); s = np.ascontiguousarray(np.repeat(s, n_maps, axis=0))
        assert m.shape == (n_maps, self.G, self.G) and s.shape == m.shape
        _capi.check(self._lib.bn_mppi_env_attach(self._h, C.c_void_p(m.ctypes.data), C.c_void_p(s.ctypes.data),
                                                 _capi.BN_MEM_HOST, goal_threshold, delta_t, seed))
        _capi.check(self._lib.bn_mppi_env_set_freeze(self._h, 1 if freeze_on_goal else 0))

    def episode(self, n_steps: int, states0, z_device_ptr: Optional[int] = None, eps_ptr: Optional[int] = None,
                kind: int = _capi.BN_NOISE_PHILOX, eps_ring: int = 1, eps_stride: int = 0, wait: bool = True):
        """n_steps closed-loop control steps on the device.  Returns (states (n+1,B,3), rewards (n,B),
        done_step (B)) when wait=True (the applied controls (n,B,2) are kept in `last_actions`); otherwise
        only enqueues (read the log later with episode_log())."""
        st = _f32(states0).reshape(self.B, 3)
        _capi.check(self._lib.bn_mppi_episode_async(self._h, n_steps, C.c_void_p(st.ctypes.data), _capi.BN_MEM_HOST,
                                                    C.c_void_p(eps_ptr), kind, eps_ring, eps_stride, C.c_void_p(z_device_ptr)))
        self._ep_steps = n_steps
        return self.episode_log() if wait else None

    def episode_log(self):
        n = self._ep_steps
        states = np.empty((n + 1, self.B, 3), np.float32)
        rewards = np.empty((n, self.B), np.float32)
        actions = np.empty((n, self.B, 2), np.float32)
        done = np.empty(self.B, np.int32)
        _capi.check(self._lib.bn_mppi_episode_log(self._h, _fp(states), _fp(rewards), _fp(actions),
                                                  done.ctypes.data_as(C.POINTER(C.c_int32))))
        self.last_actions = actions
        return states, rewards, done

    def sync(self):
        """Write the pending tail of the latest solve and wait for the stream."""
        _capi.check(self._lib.bn_mppi_sync(self._h))

    def recovery_count(self) -> int:
        """How many times batches were re-run after an expired device-side wait of an overlapped launch (0 normally)."""
        return int(self._lib.bn_mppi_recovery_count(self._h))

    def overlap_mode(self) -> int:
        """0 overlapped, 1 one stream by the handle's own choice (a co-tenant on the device), 2 one stream after an expired wait, 3 never overlaps."""
        return int(self._lib.bn_mppi_overlap_mode(self._h))

    def flush(self):
        """Enqueue the pending tail (U*, X*, weights of the latest solve) without waiting."""
        _capi.check(self._lib.bn_mppi_flush(self._h))

    # -- outputs (reference layouts) -------------------------------------------------------
    def weights(self, instance: int = 0) -> np.ndarray:
        out = np.empty(self.K, np.float32)
        _capi.check(self._lib.bn_mppi_get_weights(self._h, instance, _fp(out)))
        return out

    def costs(self, instance: int = 0) -> np.ndarray:
        out = np.empty(self.K, np.float32)
        _capi.check(self._lib.bn_mppi_get_costs(self._h, instance, _fp(out)))
        return out

    def states(self, instance: int = 0) -> np.ndarray:
        out = np.empty((self.K, self.T + 1, 3), np.float32)
        _capi.check(self._lib.bn_mppi_get_states(self._h, instance, _fp(out)))
        return out

    def controls(self, instance: int = 0) -> np.ndarray:
        out = np.empty((self.K, self.T, 2), np.float32)
        _capi.check(self._lib.bn_mppi_get_controls(self._h, instance, _fp(out)))
        return out

    def philox_noise(self, solve_index: int, instance: int = 0) -> np.ndarray:
        out = np.empty((self.K, self.T, 2), np.float32)
        _capi.check(self._lib.bn_mppi_get_philox_noise(self._h, instance, solve_index, _fp(out)))
        return out

    def top_samples(self, n: int, instance: int = 0):
        s = np.empty((n, self.T + 1, 3), np.float32)
        w = np.empty(n, np.float32)
        _capi.check(self._lib.bn_mppi_get_top_samples(self._h, instance, n, _fp(s), _fp(w)))
        return s, w

    def reroll_async_device(self, out_ptr: int, n: int, idx_ptr: Optional[int] = None, instance: int = 0):
        """Rows idx[0..n) (None: 0..n-1) of the latest solve's trajectory batch, regenerated into out (n,T+1,3) on the device."""
        _capi.check(self._lib.bn_mppi_reroll_async(self._h, instance, C.c_void_p(idx_ptr), n, C.c_void_p(out_ptr)))

    def device_buffer(self, buf_id: int):
        ptr, nbytes = C.c_void_p(), C.c_size_t()
        _capi.check(self._lib.bn_mppi_device_buffer(self._h, buf_id, C.byref(ptr), C.byref(nbytes)))
        return ptr.value, nbytes.value

    def solve_count(self) -> int:
        return int(self._lib.bn_mppi_solve_count(self._h))

    def arithmetic(self) -> str:
        """'spec' (carried heading vector, fused transit: the default) or 'reference_order' (robot_model.py:86-88 as written)."""
        return "reference_order" if self._lib.bn_mppi_arithmetic(self._h) == 1 else "spec"

    def fast_quotient(self) -> int:
        """2: power-of-two resolution (exact multiply); 1: the three-instruction quotient, validated exhaustively at create."""
        return int(self._lib.bn_mppi_fast_quotient(self._h))

    def launches_per_solve(self) -> int:
        return int(self._lib.bn_mppi_launches_per_solve(self._h))

    def kernel_ms(self):
        r, f, n = C.c_float(), C.c_float(), C.c_int32()
        _capi.check(self._lib.bn_mppi_kernel_ms(self._h, C.byref(r), C.byref(f), C.byref(n)))
        return r.value, f.value, n.value

    def algorithmic_bytes(self, injected_noise: bool = True, window: bool = False) -> int:
        """SURVEY 8d's algorithmic bytes of one solve; window=True counts the reachable LDS window instead of the whole map."""
        kind = _capi.BN_NOISE_DEVICE_KT2 if injected_noise else _capi.BN_NOISE_PHILOX
        fn = self._lib.bn_mppi_algorithmic_bytes_window if window else self._lib.bn_mppi_algorithmic_bytes
        return int(fn(self._h, kind))
