"""ctypes front end of oracle/mppi_oracle.c (TEST INFRASTRUCTURE ONLY).

The C file is the restatement; this module only marshals numpy arrays.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

TRIG_LIBM = 0
TRIG_SPEC = 1            # the spec: heading vector carried by rotation within a rollout (oracle/mppi_oracle.c)
TRIG_SPEC_PER_STEP = 2   # rounds 1-2: bn_sincos_spec of every step's heading


class OracleParams(C.Structure):
    _fields_ = [
        ("K", C.c_int32), ("T", C.c_int32), ("G", C.c_int32), ("trig", C.c_int32),
        ("res", C.c_float), ("x0", C.c_float), ("y0", C.c_float),
        ("x_lo", C.c_float), ("x_hi", C.c_float), ("y_lo", C.c_float), ("y_hi", C.c_float),
        ("dt", C.c_float), ("thr", C.c_float), ("lambda_", C.c_float),
        ("sigma", C.c_float * 2), ("inv_var", C.c_float * 2),
        ("u_min", C.c_float * 2), ("u_max", C.c_float * 2), ("goal", C.c_float * 2),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (make -C oracle)."""
    src = os.path.join(_HERE, "mppi_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        assert _lib.oracle_params_size() == C.sizeof(OracleParams)
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f32(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    if shape is not None:
        assert a.shape == tuple(shape), (a.shape, shape)
    return a


def make_params(K, T, G, res, goal, thr=0.3, lambda_=0.5, sigma=(0.5, 0.5), inv_var=None,
                u_min=(0.0, -1.0), u_max=(1.0, 1.0), dt=0.1, x_limits=None, y_limits=None,
                trig=TRIG_SPEC) -> OracleParams:
    """Geometry defaults follow grid_map.py:42-50 (limits [0, G*res], origin = lower limit)."""
    if x_limits is None:
        c = G * res / 2
        x_limits = (c - G / 2 * res, c + G / 2 * res)
    if y_limits is None:
        y_limits = x_limits
    if inv_var is None:
        s32 = np.asarray(sigma, np.float32)
        inv_var = (np.float32(1) / (s32 * s32)).tolist()
    p = OracleParams()
    p.K, p.T, p.G, p.trig = int(K), int(T), int(G), int(trig)
    p.res = res
    p.x0, p.y0 = x_limits[0], y_limits[0]
    p.x_lo, p.x_hi = x_limits
    p.y_lo, p.y_hi = y_limits
    p.dt, p.thr, p.lambda_ = dt, thr, lambda_
    for i in range(2):
        p.sigma[i] = sigma[i]
        p.inv_var[i] = inv_var[i]
        p.u_min[i] = u_min[i]
        p.u_max[i] = u_max[i]
        p.goal[i] = float(goal[i])
    return p


def solve(p: OracleParams, R, state, mean, eps):
    """One MPPI solve. Returns dict(U, X, cost, w, Ustar, Xstar) of float32 arrays."""
    K, T, G = p.K, p.T, p.G
    R = _f32(R, (G, G)); state = _f32(state, (3,)); mean = _f32(mean, (T, 2)); eps = _f32(eps, (K, T, 2))
    out = dict(U=np.empty((K, T, 2), np.float32), X=np.empty((K, T + 1, 3), np.float32),
               cost=np.empty(K, np.float32), w=np.empty(K, np.float32),
               Ustar=np.empty((T, 2), np.float32), Xstar=np.empty((T + 1, 3), np.float32))
    lib().oracle_solve(C.byref(p), _fp(R), _fp(state), _fp(mean), _fp(eps), _fp(out["U"]),
                       _fp(out["X"]), _fp(out["cost"]), _fp(out["w"]), _fp(out["Ustar"]),
                       _fp(out["Xstar"]))
    return out


def rollout(p: OracleParams, R, state, u):
    R = _f32(R, (p.G, p.G)); state = _f32(state, (3,)); u = _f32(u, (p.T, 2))
    X = np.empty((p.T + 1, 3), np.float32)
    lib().oracle_rollout(C.byref(p), _fp(R), _fp(state), _fp(u), _fp(X))
    return X


def env_step(p: OracleParams, trav, state, u):
    state = _f32(state, (3,)); u = _f32(u, (2,))
    nxt = np.empty(3, np.float32)
    lib().oracle_env_step(C.byref(p), C.c_float(trav), _fp(state), _fp(u), _fp(nxt))
    return nxt


def env_step_sampled(p: OracleParams, MU, SG, z, goal_thr, state, u):
    """PlanetaryEnv.step (planetary_env.py:189-219) with the slip draw z explicit; p.dt is the environment's delta_t.
    Returns (next_state (3,), reward, terminated)."""
    MU = _f32(MU, (p.G, p.G)); SG = _f32(SG, (p.G, p.G)); state = _f32(state, (3,)); u = _f32(u, (2,))
    nxt = np.empty(3, np.float32); rw = C.c_float(); term = C.c_int32()
    lib().oracle_env_step_sampled(C.byref(p), _fp(MU), _fp(SG), C.c_float(z), C.c_float(goal_thr), _fp(state), _fp(u), _fp(nxt),
                                  C.byref(rw), C.byref(term))
    return nxt, np.float32(rw.value), bool(term.value)


def collision_check(p: OracleParams, MU, SG, states, z, stuck_thr):
    """PlanetaryEnv.collision_check (planetary_env.py:221-232): states (..., 3), z (...) -> bool (...)."""
    MU = _f32(MU, (p.G, p.G)); SG = _f32(SG, (p.G, p.G))
    st = _f32(states); zz = _f32(z)
    assert st.shape[:-1] == zz.shape and st.shape[-1] == 3
    out = np.empty(zz.size, np.uint8)
    lib().oracle_collision_check(C.byref(p), _fp(MU), _fp(SG), _fp(st), _fp(zz), C.c_int64(zz.size), C.c_float(stuck_thr),
                                 out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out.reshape(zz.shape).astype(bool)


def sincos(x, trig=TRIG_SPEC):
    x = _f32(x).ravel()
    s = np.empty_like(x); c = np.empty_like(x)
    lib().oracle_sincos(C.c_int32(trig), C.c_int64(x.size), _fp(x), _fp(s), _fp(c))
    return s, c


def rotate(d, cs, sn):
    """(cs, sn) <- R(d)(cs, sn) by the spec's small-angle rotation; arrays of equal size. Returns new (cs, sn)."""
    d = _f32(d).ravel(); cs = _f32(cs).ravel().copy(); sn = _f32(sn).ravel().copy()
    lib().oracle_rotate(C.c_int64(d.size), _fp(d), _fp(cs), _fp(sn))
    return cs, sn


def dwa(p: OracleParams, R, state, actions, sub_goal=None):
    """DWA.forward for constant-control candidates `actions` (NA,2). Returns dict(X, cost, w, best)."""
    R = _f32(R, (p.G, p.G)); state = _f32(state, (3,)); actions = _f32(actions)
    NA = actions.shape[0]
    sg = _f32(sub_goal if sub_goal is not None else [p.goal[0], p.goal[1]], (2,))
    X = np.empty((NA, p.T + 1, 3), np.float32); cost = np.empty(NA, np.float32); w = np.empty(NA, np.float32)
    lib().oracle_dwa.restype = C.c_int32
    best = lib().oracle_dwa(C.byref(p), _fp(R), _fp(state), _fp(actions), C.c_int32(NA), _fp(sg), _fp(X), _fp(cost), _fp(w))
    return dict(X=X, cost=cost, w=w, best=int(best))


def dwa_sub_goal(p: OracleParams, R, state, action0, path, lookahead):
    """The sub-goal DWA.forward uses (dwa.py:240-244, 260-285): picked from candidate 0's aliased slot-0 state.
    Returns (sub_goal (2,), the state the rule saw (3,), index)."""
    R = _f32(R, (p.G, p.G)); state = _f32(state, (3,)); a0 = _f32(action0, (2,)); path = _f32(path)
    sel = np.empty(3, np.float32); sg = np.empty(2, np.float32)
    lib().oracle_dwa_sub_goal.restype = C.c_int32
    idx = lib().oracle_dwa_sub_goal(C.byref(p), _fp(R), _fp(state), _fp(a0), _fp(path), C.c_int32(path.shape[0]), C.c_float(lookahead),
                                    _fp(sel), _fp(sg))
    return sg, sel, int(idx)


def solve_sampled(p: OracleParams, MU, SG, state, mean, eps, zt, zc, zo):
    """MPPI solve with sampled slip (BASELINE config 3); zt (K,T), zc (K,T+1), zo (T) standard normals."""
    K, T, G = p.K, p.T, p.G
    MU = _f32(MU, (G, G)); SG = _f32(SG, (G, G)); state = _f32(state, (3,)); mean = _f32(mean, (T, 2))
    eps = _f32(eps, (K, T, 2)); zt = _f32(zt, (K, T)); zc = _f32(zc, (K, T + 1)); zo = _f32(zo, (T,))
    out = dict(U=np.empty((K, T, 2), np.float32), X=np.empty((K, T + 1, 3), np.float32), cost=np.empty(K, np.float32),
               w=np.empty(K, np.float32), Ustar=np.empty((T, 2), np.float32), Xstar=np.empty((T + 1, 3), np.float32))
    lib().oracle_solve_sampled(C.byref(p), _fp(MU), _fp(SG), _fp(state), _fp(mean), _fp(eps), _fp(zt), _fp(zc), _fp(zo),
                               _fp(out["U"]), _fp(out["X"]), _fp(out["cost"]), _fp(out["w"]), _fp(out["Ustar"]), _fp(out["Xstar"]))
    return out


def portable_normal(seed: int, stream: int, n: int):
    """n standard normals of stream `stream` under `seed`: identical bits on every host (oracle_portable_normal).
    Test-input generator for the census fixtures; not part of the parity spec."""
    out = np.empty(int(n), np.float32)
    lib().oracle_portable_normal(C.c_uint64(seed), C.c_uint64(stream), C.c_int64(n), _fp(out))
    return out
