"""MI355X-native MPPI local planner with the reference's Python interface.

Mirror of `MPPI(nn.Module)` in the reference's
src/planners/local_planners/mppi.py (constructor :23-36, forward :130-219,
get_top_samples :221-240): same argument names and meaning, same return shapes,
same assertion behaviour, so `PlanetaryEnv` loops and the tutorials can switch by
changing the import.  All arithmetic runs in hand-written HIP kernels behind the
C ABI of include/benchnav_mppi.h; torch is used only for device memory views,
stream sharing and the (optional) reference-compatible noise stream.  There is
no CPU fallback: construction raises without a gfx950 device.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import _capi


def _raw_stream(dev_index: int) -> int:
    """hipStream_t of torch's current stream on the device (forward() compares it with the planner's every call)."""
    return torch._C._cuda_getCurrentRawStream(dev_index)


if not hasattr(torch._C, "_cuda_getCurrentRawStream"):          # (a torch without the private accessor: the public one)
    def _raw_stream(dev_index: int) -> int:                     # noqa: F811
        return torch.cuda.current_stream(dev_index).cuda_stream


class _CallState:
    """Per-call state of MPPI.forward.  A plain object: nn.Module.__setattr__ inspects every assignment (parameters,
    buffers, sub-modules), which costs microseconds per attribute in a loop that runs once per control step."""
    __slots__ = ("eps", "noise_cache", "rolled", "state")

    def __init__(self):
        self.eps = self.noise_cache = self.rolled = self.state = None


class _DevArray:
    """Library-owned device memory exposed to torch through __cuda_array_interface__."""

    def __init__(self, ptr: int, shape, strides_elems=None):
        self.__cuda_array_interface__ = {
            "shape": tuple(int(s) for s in shape), "typestr": "<f4", "data": (int(ptr), False), "version": 2,
            "strides": None if strides_elems is None else tuple(int(s) * 4 for s in strides_elems),
        }


# Every private attribute of the reference's objects the native planners read, by reference class.  The golden
# fixture tests/golden/boundary.json records the attributes the REAL classes carry (tests/golden/make_golden.py imports them);
# tests/test_boundary_golden.py checks this table against it, and rebuilds reference-shaped objects from nothing but the recorded
# names to run _planner_inputs / env_inputs on them.
REFERENCE_READS = {
    "UnicycleModel": ["_grid_map", "_model_config", "_traversability_model", "min_action", "max_action"],   # robot_model.py:33-44
    "ModelConfig": ["mode"],                                                                                  # utils.py:10-14
    "TraversabilityModel": ["_risks"],                                                                        # traversability_model.py:24-26
    "GridMap": ["grid_size", "resolution", "x_limits", "y_limits", "distributions"],                          # grid_map.py:40-58
    "Objectives": ["_goal_pos", "_stuck_threshold"],                                                          # objectives.py:26-27
    "PlanetaryEnv": ["_grid_map", "_start_pos", "_goal_pos", "_delta_t", "_time_limit", "stuck_threshold",    # planetary_env.py:58-92
                     "_goal_threshold", "_seed"],
}


def _planner_inputs(dynamics, objectives, sampled_slip=False):
    """What the native planner reads from the reference-shaped `dynamics` / `objectives`
    objects: the risk map, the grid geometry, the action bounds, the goal
    and the stuck threshold.  With sampled_slip: the latent slip model instead of the risk map."""
    cfg = getattr(dynamics, "_model_config", None)
    mode = getattr(cfg, "mode", "inference")
    gm = dynamics._grid_map
    slip_std = None
    if sampled_slip:
        if mode != "observation":
            raise TypeError("sampled_slip=True plans with observation-mode dynamics (got mode %r)" % (mode,))
        latent = gm.distributions["latent_models"]              # traversability_model.py:66-68
        risks, slip_std = latent.mean, latent.stddev
    else:
        if mode != "inference":
            # the reference's transit returns a tuple in observation mode and MPPI.forward fails on it
            raise TypeError("MPPI needs dynamics in 'inference' mode (got %r); sampled_slip=True opts into planning "
                            "with the sampled slip of observation mode" % (mode,))
        risks = dynamics._traversability_model._risks
    return dict(risks=risks, slip_std=slip_std, grid_size=int(gm.grid_size), resolution=float(gm.resolution),
                x_limits=(float(gm.x_limits[0]), float(gm.x_limits[1])),
                y_limits=(float(gm.y_limits[0]), float(gm.y_limits[1])),
                goal=objectives._goal_pos, stuck_threshold=float(objectives._stuck_threshold))


class MPPI(nn.Module):
    """Model Predictive Path Integral control on one MI355X.

    Extra keyword arguments (not in the reference):
      noise  "torch_device"  (default) draw eps with torch's generator on the planner's device, as the
                      reference does when it runs on a GPU (seeded by torch.manual_seed(seed));
             "torch"  draw eps with torch's CPU generator exactly like the reference does on CPU
                      (bit-identical stream for the same seed: what the golden fixtures hold), upload it;
             "philox"        generate eps inside the rollout kernel (fastest).
      sampled_slip    plan with observation-mode dynamics (BASELINE config 3): every traversability lookup of the
                      rollouts draws slip ~ Normal(mean, std)[cell] from `grid_map.distributions["latent_models"]`
                      (traversability_model.py:65-69).  The reference's own MPPI cannot (its transit returns a tuple
                      there), so this is opt-in; without it observation-mode dynamics raise TypeError as in the reference.
      store_controls  keep `_perturbed_action_seqs` in HBM (the reference always has it).
      copy_outputs    return fresh tensors from forward() like the reference (ONE device copy of the packed U* | X*
                      block); False returns views of the planner's buffers (overwritten by the next call).
      lean            do not materialise `_state_seq_batch` (70 % of a solve's HBM bytes): get_top_samples re-rolls the
                      winners on demand and `_state_seq_batch` re-rolls all K rows when it is read, bit-identical either way.
      host_loop       opt-in for the loop of test_mppi.py:174-181 -- one forward(state) per control step with a CPU `state`, the host
                      consuming `first_action()` -- with noise="philox": every forward() also enqueues the NEXT solve's launch, which gets
                      its launch latency, noise, mean and window behind it while the host is busy and then waits on the device for the
                      state the next forward() hands over (25 -> ~14 us per control step).  Outputs stay stream-ordered.  A launch
                      that waits is cancelled by any other library call of the planner (`release()` is the cheapest) and gives up by itself
                      after ~50 ms (the next forward() then starts over with an ordinary launch); a device-wide
                      `torch.cuda.synchronize()` issued while it waits blocks that long -- call `release()` first.
                      host_loop="actions": for a loop that consumes `first_action()` and only now and then anything else -- forward()
                      leaves torch's stream unordered behind the solve (that stream-wait is 3-5 us of every control step's host time);
                      the planner's own attributes and methods (`_weights`, `_state_seq_batch`, `get_top_samples` ...) make up for it
                      when they are read, the tensors forward() RETURNED are valid for torch work enqueued after `order_outputs()`.
      reference_order the transit in the reference's own operation order (robot_model.py:86-88: sin / cos of every step's
                      heading, x + ((trav v) cos) dt): no cell flips against the reference beyond what libm vs SLEEF gives
                      (DESIGN.md 5), on the same kernels (one launch per solve) at ~1.4x the single-instance latency -- the chain's extra
                      instructions; 1-8 % for batched launches.  The default arithmetic leaves about one
                      rollout in 30 000 (T = 50) in a neighbouring cell.  `arithmetic` tells which one a planner runs; a
                      `delta_t * max|omega|` above 0.5 rad selects the reference order by itself.

    Streams: the planner enqueues on the torch stream that was current for its device at construction.  forward() called
    under another current stream fences the two with events (correct, slower); keep one stream for the best latency.
    """

    def __init__(self, horizon: int, num_samples: int, dim_state: int, dim_control: int, dynamics, objectives,
                 sigmas: torch.Tensor, lambda_: float, device=torch.device("cuda"), dtype=torch.float32,
                 seed: int = 42, *, noise: str = "torch_device", store_controls: bool = True,
                 copy_outputs: bool = True, profile: bool = False, delta_t: float = 0.1, sampled_slip: bool = False,
                 lean: bool = False, reference_order: bool = False, host_loop=False) -> None:
        super().__init__()
        torch.manual_seed(seed)                                    # mppi.py:55

        assert dynamics.min_action.shape == (dim_control,), "minimum actions must be a tensor of shape (dim_control,)"
        assert dynamics.max_action.shape == (dim_control,), "maximum actions must be a tensor of shape (dim_control,)"
        assert sigmas.shape == (dim_control,), "sigmas must be a tensor of shape (dim_control,)"
        if dim_state != 3 or dim_control != 2:
            raise ValueError("the native planner implements the unicycle model: dim_state=3, dim_control=2")
        if dtype != torch.float32:
            raise ValueError("the native planner computes in float32")
        if noise not in ("torch", "torch_device", "philox"):
            raise ValueError(f"unknown noise mode {noise!r}")
        if not torch.cuda.is_available():
            raise RuntimeError("benchnav_amd.MPPI needs an MI355X (gfx950) device; there is no CPU fallback")
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError(f"benchnav_amd.MPPI runs on the GPU only (device={device!r})")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self._device = dev
        self._dtype = dtype

        self._horizon = horizon
        self._num_samples = num_samples
        self._dim_state = dim_state
        self._dim_control = dim_control
        self._dynamics = dynamics
        self._stage_cost = objectives.stage_cost
        self._terminal_cost = objectives.terminal_cost
        self._u_min = dynamics.min_action.clone().detach().to(dev, dtype)
        self._u_max = dynamics.max_action.clone().detach().to(dev, dtype)
        self._sigmas = sigmas.clone().detach().to(dev, dtype)
        self._lambda = lambda_
        self._noise_mode = noise
        self._copy_outputs = copy_outputs
        # inverse covariance computed the reference's way (mppi.py:94-97)
        cov = torch.diag(sigmas.detach().cpu().to(dtype) ** 2)
        inv_cov = torch.inverse(cov)
        self._inv_covariance = inv_cov.to(dev, dtype)
        self._sample_shape = torch.Size([num_samples, horizon])

        inp = _planner_inputs(dynamics, objectives, sampled_slip)
        lib = _capi.load()
        cfg = _capi.Config()
        lib.bn_mppi_config_init(C.byref(cfg))
        cfg.device_id = dev.index
        cfg.horizon, cfg.num_samples, cfg.num_instances = horizon, num_samples, 1
        cfg.grid_size, cfg.resolution = inp["grid_size"], inp["resolution"]
        s_cpu = sigmas.detach().cpu().to(dtype)
        umin, umax = dynamics.min_action.detach().cpu().to(dtype), dynamics.max_action.detach().cpu().to(dtype)
        for i in range(2):
            cfg.x_limits[i], cfg.y_limits[i] = inp["x_limits"][i], inp["y_limits"][i]
            cfg.sigma[i], cfg.inv_var[i] = float(s_cpu[i]), float(inv_cov[i, i])
            cfg.u_min[i], cfg.u_max[i] = float(umin[i]), float(umax[i])
        cfg.lambda_, cfg.dt, cfg.stuck_threshold = lambda_, delta_t, inp["stuck_threshold"]
        cfg.seed = seed
        cfg.flags = ((_capi.BN_FLAG_STORE_CONTROLS if store_controls else 0) | (_capi.BN_FLAG_PROFILE if profile else 0)
                     | (_capi.BN_FLAG_SAMPLED_SLIP if sampled_slip else 0) | (_capi.BN_FLAG_LEAN if lean else 0)
                     | (_capi.BN_FLAG_REFERENCE_ORDER if reference_order else 0)
                     | (_capi.BN_FLAG_HOST_PACED if (host_loop and noise == "philox") else 0)
                     | (_capi.BN_FLAG_UNORDERED_OUTPUTS if (host_loop == "actions" and noise == "philox") else 0))
        self._lean = bool(lean)
        with torch.cuda.device(dev):
            self._stream = torch.cuda.current_stream(dev)
            cfg.stream = self._stream.cuda_stream
            self._stream_id, self._dev_index = self._stream.cuda_stream, dev.index
            self._handle = C.c_void_p()
            _capi.check(lib.bn_mppi_create(C.byref(cfg), C.byref(self._handle)))
        self._lib = lib
        risks = inp["risks"].detach().to(torch.float32).contiguous()
        assert risks.shape == (inp["grid_size"], inp["grid_size"])
        self.set_risk_map(risks)
        if sampled_slip:
            std = inp["slip_std"].detach().to(torch.float32).contiguous()
            assert std.shape == risks.shape
            self._slip_std = std.to(dev)
            _capi.check(lib.bn_mppi_set_slip_std(self._handle, 0, C.c_void_p(self._slip_std.data_ptr()), _capi.BN_MEM_DEVICE))
        self.set_goal(inp["goal"])

        K, T = num_samples, horizon
        Kp = int(lib.bn_mppi_row_pitch(self._handle))        # rows are pitched to 64*ceil(K/64) floats
        self._buf_X = None if lean else self._wrap(_capi.BN_BUF_STATES, (T + 1, 3, Kp))[:, :, :K]
        self._host_loop = int(lib.bn_mppi_host_paced(self._handle)) >= 1
        self._unordered = self._host_loop and host_loop == "actions"
        # host-paced solves alternate between two trajectory / control buffers (two launches in flight): bn_mppi_states_buffer_index
        self.__dict__["_buf_X_alt"] = self._wrap(_capi.BN_BUF_STATES_ALT, (T + 1, 3, Kp))[:, :, :K] if (self._host_loop and not lean) else None
        self._buf_w = self._wrap(_capi.BN_BUF_WEIGHTS, (K,))
        self._buf_cost = self._wrap(_capi.BN_BUF_COSTS, (K,))
        self._buf_U = self._wrap(_capi.BN_BUF_CONTROLS, (T, 2, Kp))[:, :, :K] if store_controls else None
        self.__dict__["_buf_U_alt"] = self._wrap(_capi.BN_BUF_CONTROLS_ALT, (T, 2, Kp))[:, :, :K] if (self._host_loop and store_controls) else None
        self._buf_out = self._wrap(_capi.BN_BUF_USTAR_XSTAR, (T * 2 + (T + 1) * 3,))     # U* | X*, one block
        self._buf_ustar = self._buf_out[:T * 2].view(T, 2)
        self._buf_xstar = self._buf_out[T * 2:].view(1, T + 1, 3)
        self._n_out = T * 2 + (T + 1) * 3
        self._fwd = lib.bn_mppi_forward_async                   # bound once: forward() is a host hot loop
        self._fwd_state = lib.bn_mppi_forward_state_async       # ... for a state that lives on the host (test_mppi.py:174-181)
        self._h = self._handle.value

        self._buf_mean = self._wrap(_capi.BN_BUF_MEAN, (T, 2))
        self._cs = _CallState()
        self._n_out4 = self._n_out * 4
        self._fast_ok = noise == "philox" and bool(copy_outputs) and dtype == torch.float32        # forward()'s short path (a CPU state handed over by value)
        self.__dict__["_first_action_buf"] = None        # (plain attributes: nn.Module.__setattr__ is slow)
        self.__dict__["_first_action_ptr"] = None

        if noise == "torch":
            # the reference constructor consumes one (K,T,2) draw of the global CPU stream (mppi.py:105-107)
            self._cs.noise_cache = torch.empty(K, T, 2).normal_() * s_cpu
        elif noise == "torch_device":
            self._cs.noise_cache = torch.randn(K, T, 2, device=dev) * self._sigmas

    # -- plumbing ------------------------------------------------------------------
    def _wrap(self, buf_id, shape):
        ptr, nbytes = C.c_void_p(), C.c_size_t()
        _capi.check(self._lib.bn_mppi_device_buffer(self._handle, buf_id, C.byref(ptr), C.byref(nbytes)))
        n = 1
        for s in shape:
            n *= s
        assert n * 4 == nbytes.value, (shape, nbytes.value)
        return torch.as_tensor(_DevArray(ptr.value, shape), device=self._device)

    def __del__(self):
        try:                                   # also runs during interpreter shutdown, when module globals are gone
            h = self.__dict__.get("_handle")
            if h is not None and h.value:
                self._lib.bn_mppi_destroy(h)
                h.value = None
        except Exception:
            pass

    @property
    def arithmetic(self) -> str:
        """'spec' (default: carried heading vector, fused transit) or 'reference_order' (robot_model.py:86-88 as written)."""
        return "reference_order" if self._lib.bn_mppi_arithmetic(self._handle) == 1 else "spec"

    def set_risk_map(self, risks: torch.Tensor) -> None:
        """Replace the risk map (dynamics._traversability_model._risks, (G,G) [iy,ix])."""
        r = risks.detach().to(torch.float32).contiguous()
        where = _capi.BN_MEM_DEVICE if r.is_cuda else _capi.BN_MEM_HOST
        if r.is_cuda:
            torch.cuda.current_stream(r.device).synchronize()
        _capi.check(self._lib.bn_mppi_set_map(self._handle, 0, C.c_void_p(r.data_ptr()), where))

    def set_goal(self, goal_pos) -> None:
        g = torch.as_tensor(goal_pos).detach().to("cpu", torch.float32).contiguous()   # int64 goals promote (test_mppi.py:133)
        assert g.shape == (2,)
        _capi.check(self._lib.bn_mppi_set_goal(self._handle, 0, C.cast(g.data_ptr(), C.POINTER(C.c_float))))

    # -- reference attributes --------------------------------------------------------
    @property
    def _previous_action_seq(self) -> torch.Tensor:
        self._unordered and self.order_outputs()
        return self._buf_mean

    @_previous_action_seq.setter
    def _previous_action_seq(self, value: torch.Tensor) -> None:
        _capi.check(self._lib.bn_mppi_flush(self._handle))            # the mean buffer is authoritative after a flush
        self._buf_mean.copy_(torch.as_tensor(value).to(self._device, self._dtype))

    @property
    def _state_seq_batch(self) -> torch.Tensor:
        """(K, T+1, 3): a view of the planner-native (T+1, 3, K) buffer (no copy); in lean mode the rows are re-rolled
        on first access after a forward() (bit-identical to what a full-API solve stores)."""
        self._unordered and self.order_outputs()
        if self._lean:
            if self._cs.rolled is None:
                self._cs.rolled = self._reroll(None, self._num_samples)
            return self._cs.rolled
        if self._host_loop and self._lib.bn_mppi_states_buffer_index(self._h) == 1:
            return self._buf_X_alt.permute(2, 0, 1)
        return self._buf_X.permute(2, 0, 1)

    def order_outputs(self) -> None:
        """host_loop="actions": order torch's stream behind the latest solve -- before torch work that consumes the tensors forward()
        returned.  A no-op otherwise; the loop keeps its pace (the launch that waits for the next state is left alone)."""
        _capi.check(self._lib.bn_mppi_order_outputs(self._handle))

    def release(self) -> None:
        """End a `host_loop` loop: cancels the launch that waits on the device for the next state (a word in pinned memory, no
        synchronisation) and enqueues anything pending.  Every other method of the planner but forward() / first_action() does the same."""
        _capi.check(self._lib.bn_mppi_flush(self._handle))

    def _reroll(self, idx: Optional[torch.Tensor], n: int) -> torch.Tensor:
        out = torch.empty(n, self._horizon + 1, 3, device=self._device, dtype=self._dtype)
        with self._on_planner_stream():
            _capi.check(self._lib.bn_mppi_reroll_async(self._handle, 0, None if idx is None else idx.data_ptr(), n, out.data_ptr()))
        return out

    @property
    def _weights(self) -> torch.Tensor:
        self._unordered and self.order_outputs()
        return self._buf_w

    @property
    def _costs(self) -> torch.Tensor:
        self._unordered and self.order_outputs()
        return self._buf_cost

    @property
    def _perturbed_action_seqs(self) -> torch.Tensor:
        if self._buf_U is None:
            raise AttributeError("_perturbed_action_seqs is not stored (store_controls=False)")
        self._unordered and self.order_outputs()
        if self._host_loop and self._lib.bn_mppi_states_buffer_index(self._h) == 1:
            return self._buf_U_alt.permute(2, 0, 1)
        return self._buf_U.permute(2, 0, 1)

    # -- the solve ---------------------------------------------------------------------
    class _Fence:
        """Orders torch's current stream and the planner's stream around a library call when they differ."""

        def __init__(self, owner):
            self.o = owner
            self.cur = torch.cuda.current_stream(owner._device)
            self.same = self.cur.cuda_stream == owner._stream.cuda_stream

        def __enter__(self):
            if not self.same:
                self.o._stream.wait_stream(self.cur)         # inputs produced on the current stream
            return self

        def __exit__(self, *exc):
            if not self.same:
                self.cur.wait_stream(self.o._stream)         # outputs consumed on the current stream

    def _on_planner_stream(self):
        return MPPI._Fence(self)

    _OUT_POOL = 32      # fresh output tensors are carved from blocks of this many (one torch.empty per 32 forwards)

    def _alloc_out_block(self):
        """(block, U views, X views, base pointer, event) for _OUT_POOL forwards: ONE allocation, views made in bulk (the reference returns new
        tensors from every forward(); a torch.empty and two as_strided per call were 5 of the drop-in step's microseconds).  host_loop:
        the launches that will write the block run on a stream of the library's own, and the memory torch's allocator hands out may have
        been freed with reads still queued on torch's stream (ordered for work on THAT stream only): an event recorded now covers them."""
        n, T, no = self._OUT_POOL, self._horizon, self._n_out
        blk = torch.empty(n, no, device=self._device, dtype=self._dtype)
        ev = None
        if self._host_loop:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self._device))
        return blk[:, :2 * T].view(n, T, 2), blk[:, 2 * T:].view(n, 1, T + 1, 3), blk.data_ptr(), ev

    def _new_out_block(self):
        """Switch to the next output block -- allocated half a block ago (host_loop: its event has long fired by now, no wait)."""
        d = self.__dict__
        nxt = d.pop("_out_next", None) or self._alloc_out_block()
        if nxt[3] is not None and not nxt[3].query():
            nxt[3].synchronize()
        d["_out_U"], d["_out_X"], d["_out_ptr"] = nxt[0], nxt[1], nxt[2]
        d["_out_i"] = 0

    def forward(self, state: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Solve the optimal control problem (mppi.py:130-219).

        Returns (optimal_action_seq (T,2), optimal_state_seq (1,T+1,3)) on the planner's
        device, stream-ordered like any torch op.  `state` is not modified.
        """
        d = self.__dict__
        if d["_fast_ok"] and state.__class__ is torch.Tensor and state.is_cpu and state.dtype is torch.float32 and state.dim() == 1 and state.shape[0] == 3 \
                and state.is_contiguous() and _raw_stream(d["_dev_index"]) == d["_stream_id"]:
            # the host loop's call (a float32 CPU state, in-kernel noise, the planner's own stream): the state is handed over first, the
            # bookkeeping follows while the GPU works -- every microsecond in front of the C call is a microsecond of the control step
            i = d.get("_out_i", self._OUT_POOL)
            if i >= self._OUT_POOL:
                self._new_out_block()
                i = 0
            rc = d["_fwd_state"](d["_h"], state.data_ptr(), None, 0, d["_out_ptr"] + i * d["_n_out4"])
            if rc:
                _capi.check(rc)
            d["_out_i"] = i + 1
            if i == self._OUT_POOL // 2:
                d["_out_next"] = self._alloc_out_block()
            cs = d["_cs"]
            cs.eps = cs.noise_cache = cs.rolled = None
            cs.state = state
            return d["_out_U"][i], d["_out_X"][i]
        if not torch.is_tensor(state):
            state = torch.tensor(state, dtype=self._dtype)
        assert state.shape == (self._dim_state,)
        fwd = self._fwd
        if state.is_cpu:
            # a host loop's state (the reference moves it, mppi.py:140-144): taken by value at the call -- no upload, no device tensor
            if state.dtype != self._dtype or not state.is_contiguous():
                state = state.detach().to(self._dtype).contiguous()
            fwd = self._fwd_state
        elif state.device != self._device or state.dtype != self._dtype or not state.is_contiguous():
            state = state.detach().to(self._device, self._dtype).contiguous()
        mode = self._noise_mode
        if mode == "philox":
            eps, eptr, kind = None, None, _capi.BN_NOISE_PHILOX
        else:
            K, T = self._num_samples, self._horizon
            if mode == "torch":
                eps = torch.empty(K, T, 2).normal_().to(self._device)      # global CPU stream, mppi.py:149-151
            else:
                eps = torch.randn(K, T, 2, device=self._device)
            eptr, kind = eps.data_ptr(), _capi.BN_NOISE_DEVICE_KT2
        cs = self._cs
        cs.eps, cs.noise_cache, cs.rolled = eps, None, None           # _action_noises and lean rows are derived on demand
        cs.state = state                                               # keep alive until the kernels ran
        # fresh output tensors like the reference's: the tail writes its packed U* | X* block into the caller's block as well (no extra launch)
        d = self.__dict__
        i = -1
        optr = None
        if self._copy_outputs:
            i = d.get("_out_i", self._OUT_POOL)
            if i >= self._OUT_POOL:
                self._new_out_block()
                i = 0
            d["_out_i"] = i + 1
            optr = d["_out_ptr"] + i * self._n_out * 4
            if i == self._OUT_POOL // 2:
                d["_out_next"] = self._alloc_out_block()
        if _raw_stream(self._dev_index) == self._stream_id:
            rc = fwd(self._h, state.data_ptr(), eptr, kind, optr)   # solve + tail: U*, X*, weights of THIS solve, stream-ordered
        else:
            with self._on_planner_stream():
                rc = fwd(self._h, state.data_ptr(), eptr, kind, optr)
        if rc:
            _capi.check(rc)
        if i >= 0:
            return d["_out_U"][i], d["_out_X"][i]
        return self._buf_ustar, self._buf_xstar

    solve = forward

    def first_action(self) -> torch.Tensor:
        """optimal_action_seq[0] of the latest forward() as a CPU tensor (2,), as early as it exists: the tail of the solve posts it
        to pinned host memory right after the softmin merge, before it rolls out X* -- for a loop whose environment lives on the
        host (test_mppi.py:181 hands action_seq[0, :] to env.step).  Same value as `optimal_action_seq[0].cpu()`, without the
        stream synchronisation and the copy.  The returned tensor is reused by the next call."""
        fa = self._first_action_buf
        if fa is None:                                   # one CPU tensor, refilled by every call (clone it to keep a value)
            fa = self.__dict__["_first_action_buf"] = torch.empty(2, dtype=self._dtype)
            self.__dict__["_first_action_ptr"] = C.cast(fa.data_ptr(), C.POINTER(C.c_float))
        rc = self._lib.bn_mppi_first_action(self._h, 0, self._first_action_ptr)
        if rc:
            _capi.check(rc)
        return fa

    @property
    def _action_noises(self) -> Optional[torch.Tensor]:
        """The reference's `_action_noises` (eps * sigma, mppi.py:149-151) of the latest forward(); None with
        in-kernel noise.  Derived on demand: the planner consumes eps itself."""
        cs = self._cs
        if cs.noise_cache is None and cs.eps is not None:
            cs.noise_cache = cs.eps * self._sigmas
        return cs.noise_cache

    def solve_with_noise(self, state: torch.Tensor, eps: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """forward() with caller-supplied standard-normal noise eps (K,T,2) (teacher-forced parity tests)."""
        assert eps.shape == (self._num_samples, self._horizon, 2)
        st = torch.as_tensor(state).detach().to(self._device, self._dtype).contiguous()
        cs = self._cs
        cs.eps = eps.detach().to(self._device, self._dtype).contiguous()
        cs.noise_cache, cs.rolled, cs.state = None, None, st
        with self._on_planner_stream():
            _capi.check(self._fwd(self._h, st.data_ptr(), cs.eps.data_ptr(), _capi.BN_NOISE_DEVICE_KT2, None))
        return self._buf_ustar.clone(), self._buf_xstar.clone()

    def get_top_samples(self, num_samples: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """The `num_samples` highest-weight rollouts, by weight descending (mppi.py:221-240)."""
        assert num_samples <= self._num_samples
        top_indices = torch.topk(self._weights, num_samples).indices
        if self._lean and self._cs.rolled is None:                     # lean: only the winners are re-rolled
            top_samples = self._reroll(top_indices.to(torch.int32), num_samples)
        else:
            top_samples = self._state_seq_batch[top_indices]
        top_weights = self._weights[top_indices]
        order = torch.argsort(top_weights, descending=True)
        return top_samples[order], top_weights[order]

    # -- measurement helpers -------------------------------------------------------------
    def kernel_ms(self):
        r, f, n = C.c_float(), C.c_float(), C.c_int32()
        _capi.check(self._lib.bn_mppi_kernel_ms(self._handle, C.byref(r), C.byref(f), C.byref(n)))
        return r.value, f.value, n.value

    def algorithmic_bytes(self) -> int:
        kind = _capi.BN_NOISE_PHILOX if self._noise_mode == "philox" else _capi.BN_NOISE_DEVICE_KT2
        return int(self._lib.bn_mppi_algorithmic_bytes(self._handle, kind))
