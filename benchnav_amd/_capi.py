"""ctypes binding of include/benchnav_mppi.h (the C-ABI drop-in boundary).

No CPU fallback: importing works anywhere (so host logic is testable), but
`load()` raises if libbenchnav_mppi.so is missing and cannot be built, and
`bn_mppi_create` fails without a gfx950 device.
"""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

# ---- enums (include/benchnav_mppi.h) ---------------------------------------------
BN_OK = 0
BN_ERR_INVALID, BN_ERR_HIP, BN_ERR_NO_DEVICE, BN_ERR_STATE = -1, -2, -3, -4
BN_NOISE_PHILOX, BN_NOISE_HOST_KT2, BN_NOISE_DEVICE_KT2, BN_NOISE_DEVICE_T2K = 0, 1, 2, 3
BN_MEM_HOST, BN_MEM_DEVICE = 0, 1
(BN_BUF_STATES, BN_BUF_WEIGHTS, BN_BUF_COSTS, BN_BUF_CONTROLS, BN_BUF_USTAR, BN_BUF_XSTAR,
 BN_BUF_MEAN, BN_BUF_MAP, BN_BUF_GOAL, BN_BUF_USTAR_XSTAR) = range(10)
BN_FLAG_STORE_CONTROLS, BN_FLAG_SHARED_MAP, BN_FLAG_NO_LDS_WINDOW, BN_FLAG_PROFILE, BN_FLAG_PRIVATE_STREAM = 1, 2, 4, 8, 16
BN_FLAG_NO_PIPELINE = 32
BN_FLAG_SAMPLED_SLIP = 64
BN_FLAG_WAVE_KERNEL = 128
BN_FLAG_ROLE_KERNEL = 256
BN_FLAG_LEAN = 512
BN_FLAG_LAT_KERNEL = 1024
BN_FLAG_NO_OVERLAP = 2048
BN_FLAG_REFERENCE_ORDER = 4096
BN_FLAG_HOST_PACED = 8192
BN_FLAG_UNORDERED_OUTPUTS = 16384
BN_BUF_STATES_ALT, BN_BUF_CONTROLS_ALT = 10, 11
BN_RISK_EXPECTED, BN_RISK_VAR, BN_RISK_CVAR = 0, 1, 2
ABI_VERSION = 4


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("device_id", C.c_int32), ("horizon", C.c_int32),
        ("num_samples", C.c_int32), ("num_instances", C.c_int32), ("grid_size", C.c_int32),
        ("resolution", C.c_float), ("x_limits", C.c_float * 2), ("y_limits", C.c_float * 2),
        ("sigma", C.c_float * 2), ("inv_var", C.c_float * 2), ("lambda_", C.c_float),
        ("u_min", C.c_float * 2), ("u_max", C.c_float * 2), ("dt", C.c_float),
        ("stuck_threshold", C.c_float), ("seed", C.c_uint64), ("flags", C.c_uint32),
        ("stream", C.c_void_p),
    ]


# every symbol include/benchnav_mppi.h declares: name -> (restype, argtypes)
_FP = C.POINTER(C.c_float)
_H = C.c_void_p
SYMBOLS = {
    "bn_mppi_config_init": (None, [C.POINTER(Config)]),
    "bn_mppi_create": (C.c_int, [C.POINTER(Config), C.POINTER(_H)]),
    "bn_mppi_destroy": (None, [_H]),
    "bn_mppi_set_map": (C.c_int, [_H, C.c_int32, C.c_void_p, C.c_int]),
    "bn_mppi_set_slip_std": (C.c_int, [_H, C.c_int32, C.c_void_p, C.c_int]),
    "bn_mppi_get_slip_noise": (C.c_int, [_H, C.c_int32, C.c_uint64, _FP, _FP, _FP]),
    "bn_mppi_set_slip_noise": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bn_mppi_set_goal": (C.c_int, [_H, C.c_int32, _FP]),
    "bn_mppi_set_mean": (C.c_int, [_H, C.c_int32, _FP]),
    "bn_mppi_get_mean": (C.c_int, [_H, C.c_int32, _FP]),
    "bn_mppi_solve": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_void_p, C.c_int, _FP, _FP]),
    "bn_mppi_solve_async": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "bn_mppi_forward_async": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "bn_mppi_forward_state_async": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "bn_mppi_solve_n_async": (C.c_int, [_H, C.c_int32, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int32, C.c_int64]),
    "bn_mppi_set_rollout_offset": (C.c_int, [_H, C.c_int64]),
    "bn_mppi_shard_rollout_async": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "bn_mppi_shard_partials": (C.c_int, [_H, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "bn_mppi_shard_finish_async": (C.c_int, [_H, C.c_void_p, C.c_int32]),
    "bn_dist_unique_id": (C.c_int, [C.c_void_p]),
    "bn_mppi_shard_comm_prepare": (C.c_int, [_H, C.c_int32, C.c_int32]),
    "bn_mppi_shard_comm_init": (C.c_int, [_H, C.c_void_p, C.c_int32, C.c_int32]),
    "bn_mppi_shard_solve_async": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "bn_mppi_env_attach": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_uint64]),
    "bn_mppi_env_set_freeze": (C.c_int, [_H, C.c_int32]),
    "bn_mppi_env_step": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    "bn_mppi_env_collision_check": (C.c_int, [_H, C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_uint64, C.c_void_p]),
    "bn_mppi_episode_async": (C.c_int, [_H, C.c_int32, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int32, C.c_int64, C.c_void_p]),
    "bn_mppi_episode_log": (C.c_int, [_H, _FP, _FP, _FP, C.POINTER(C.c_int32)]),
    "bn_mppi_dwa_solve": (C.c_int, [_H, _FP, _FP, C.c_int32, _FP, _FP, _FP, _FP, _FP, _FP, C.POINTER(C.c_int32)]),
    "bn_mppi_dwa_forward_async": (C.c_int, [_H, C.c_void_p, C.c_void_p, _FP, C.c_float, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_float, C.c_void_p]),
    "bn_mppi_dwa_candidates": (C.c_int, [_H, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "bn_mppi_dwa_buffers": (C.c_int, [_H, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "bn_mppi_sync": (C.c_int, [_H]),
    "bn_mppi_recovery_count": (C.c_uint64, [_H]),
    "bn_mppi_overlap_mode": (C.c_int32, [_H]),
    "bn_mppi_debug_cadence": (C.c_int, [_H, C.c_int32, C.c_double]),
    "bn_mppi_first_action": (C.c_int, [_H, C.c_int32, _FP]),
    "bn_mppi_debug_expire_wait": (C.c_int, [_H]),
    "bn_mppi_flush": (C.c_int, [_H]),
    "bn_mppi_debug_host_paced": (C.c_int, [_H, C.c_int32, C.c_int32]),
    "bn_mppi_get_weights": (C.c_int, [_H, C.c_int32, _FP]),
    "bn_mppi_get_costs": (C.c_int, [_H, C.c_int32, _FP]),
    "bn_mppi_get_states": (C.c_int, [_H, C.c_int32, _FP]),
    "bn_mppi_get_controls": (C.c_int, [_H, C.c_int32, _FP]),
    "bn_mppi_get_philox_noise": (C.c_int, [_H, C.c_int32, C.c_uint64, _FP]),
    "bn_mppi_get_top_samples": (C.c_int, [_H, C.c_int32, C.c_int32, _FP, _FP]),
    "bn_mppi_reroll_async": (C.c_int, [_H, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "bn_mppi_device_buffer": (C.c_int, [_H, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "bn_mppi_solve_count": (C.c_uint64, [_H]),
    "bn_mppi_arithmetic": (C.c_int32, [_H]),
    "bn_mppi_launches_per_solve": (C.c_int32, [_H]),
    "bn_mppi_launches_per_forward": (C.c_int32, [_H]),
    "bn_mppi_host_paced": (C.c_int32, [_H]),
    "bn_mppi_order_outputs": (C.c_int, [_H]),
    "bn_mppi_states_buffer_index": (C.c_int32, [_H]),
    "bn_mppi_fast_quotient": (C.c_int32, [_H]),
    "bn_mppi_row_pitch": (C.c_int32, [_H]),
    "bn_mppi_kernel_ms": (C.c_int, [_H, _FP, _FP, C.POINTER(C.c_int32)]),
    "bn_mppi_algorithmic_bytes": (C.c_int64, [_H, C.c_int]),
    "bn_mppi_algorithmic_bytes_window": (C.c_int64, [_H, C.c_int]),
    "bn_risk_map_infer": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int32, C.c_int, C.c_float,
                                    C.c_int32, C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_int]),
    "bn_risk_last_error": (C.c_char_p, []),
    "bn_device_math_eval": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "bn_last_error": (C.c_char_p, []),
    "bn_mppi_abi_version": (C.c_int, []),
}

_lib = None


class BenchnavError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"benchnav_mppi error {code}: {msg}")
        self.code = code


def lib_path() -> str:
    return _build.LIB_PATH


def _preload_hip_runtime():
    """Make sure ONE HIP/HSA runtime serves both this library and torch.

    torch wheels bundle their own libamdhip64.so (soname libamdhip64.so.7) and load it by
    file name; our library needs `libamdhip64.so.7`.  If ours pulled in /opt/rocm's copy
    first, a later `import torch` would load a second runtime, which then sees no GPU.
    Loading torch's copy first satisfies both by soname.  Without torch installed the
    system ROCm runtime is used.
    """
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def _open_checked(path):
    """dlopen + bind every symbol of SYMBOLS.  Returns (lib, None) or (lib, what does not match)."""
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name, None)
        if fn is None:
            return lib, f"symbol {name} missing: the header and the library disagree"
        fn.restype = res
        fn.argtypes = args
    if lib.bn_mppi_abi_version() != ABI_VERSION:
        return lib, f"ABI version {lib.bn_mppi_abi_version()} != binding {ABI_VERSION}"
    return lib, None


def load(build_if_missing: bool = True):
    """dlopen the in-tree library (building it with hipcc first if it is absent)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    _preload_hip_runtime()
    # A missing library is built.  A library older than its sources is rebuilt only on request (BENCHNAV_REBUILD_IF_STALE=1):
    # file times do not survive every way a tree gets copied to a GPU box, and a spurious 60 s rebuild in every test
    # process would be worse than the stale-binary risk.  A library that does not MATCH this binding -- another ABI version,
    # or a symbol of include/benchnav_mppi.h missing -- is rebuilt once, whatever its file time says.
    stale = os.environ.get("BENCHNAV_REBUILD_IF_STALE") == "1" and _build.is_stale()
    if not os.path.exists(path) or stale:
        if not build_if_missing:
            raise RuntimeError(f"{path} is missing; run `python -m benchnav_amd.build`")
        _build.build_library()
    lib, why = _open_checked(path)
    if why and build_if_missing:
        import _ctypes
        _ctypes.dlclose(lib._handle)     # or the next dlopen of this path returns the old image
        del lib
        _build.build_library(force=True)
        lib, why = _open_checked(path)
    if why:
        raise RuntimeError(f"{path} does not match this binding ({why}); run `python -m benchnav_amd.build`")
    _lib = lib
    return lib


def check(code: int):
    if code != BN_OK:
        raise BenchnavError(code, load().bn_last_error().decode("utf-8", "replace"))
