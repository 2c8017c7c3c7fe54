"""In-tree build of the gfx950 library (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_PKG, "csrc")
LIB_DIR = os.path.join(_PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libbenchnav_mppi.so")
# one translation unit per kernel family: they compile in parallel (the role kernel alone is 72 template instances)
SOURCES = ["rollout_role_philox.hip", "rollout_role_kt2.hip", "rollout_role_t2k.hip", "rollout_role_ref_philox.hip", "rollout_role_ref_kt2.hip", "rollout_role_ref_t2k.hip", "rollout_wave.hip", "rollout_wave_ref.hip", "rollout_sampled.hip",
           "rollout_lat_host.hip", "rollout_lat_host_ref.hip",
           "rollout_lat_self_philox.hip", "rollout_lat_self_kt2.hip", "rollout_lat_self_t2k.hip",
           "rollout_lat_self_ref_philox.hip", "rollout_lat_self_ref_kt2.hip", "rollout_lat_self_ref_t2k.hip",
           "mppi_kernels.hip", "mppi_capi.cpp", "risk_kernels.hip"]
# per-source flags.  rollout_wave_ref.hip: see the note at its top (a register-allocation fault behind the SLP vectoriser)
EXTRA_FLAGS = {f: ["-fno-slp-vectorize"] for f in ("rollout_wave_ref.hip", "rollout_role_ref_philox.hip", "rollout_role_ref_kt2.hip", "rollout_role_ref_t2k.hip",
                                                      "rollout_lat_host_ref.hip", "rollout_lat_self_ref_philox.hip", "rollout_lat_self_ref_kt2.hip",
                                                      "rollout_lat_self_ref_t2k.hip")}
# every header / include file under csrc/ (globbed: a new .inc cannot be forgotten here) + the public header
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))) + [os.path.join("..", "..", "include", "benchnav_mppi.h")]

# -ffp-contract=off: the arithmetic spec fixes where FMAs are (explicit __builtin_fmaf only).
# Division and sqrt stay correctly rounded (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
               "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MPPI planner needs the ROCm toolchain to build")
    return exe


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    """Compile csrc/*.hip|cpp into lib/libbenchnav_mppi.so for gfx950.

    Safe under concurrent callers (every rank of a torchrun job importing the package at once): one holds an exclusive file
    lock and builds into private object / library names, then publishes with an atomic rename; the others wait for the lock,
    find the library fresh and return."""
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    import fcntl
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not is_stale():         # another process built it while this one waited
                return LIB_PATH
            return _build_locked(verbose, extra_flags)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose, extra_flags) -> str:
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    compile_flags = [f for f in HIPCC_FLAGS if f != "-shared"]
    jobs = []
    for src in SOURCES:
        obj = os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc(), *compile_flags, *EXTRA_FLAGS.get(src, []), *extra_flags, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        jobs.append((src, obj, subprocess.Popen(cmd)))
    objs = []
    for src, obj, proc in jobs:
        if proc.wait() != 0:
            for _, _, other in jobs:
                other.wait()
            raise RuntimeError(f"hipcc failed on {src}")
        objs.append(obj)
    tmp = LIB_PATH + f".tmp{os.getpid()}"
    link = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", tmp]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    os.replace(tmp, LIB_PATH)                       # readers see the old library or the new one, never a half-written file
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
