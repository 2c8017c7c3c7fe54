"""Differential sweeps (GPU): knobs that must NOT change a result.  Library of tests/test_gpu_differential.py and of the wide one-off
sweeps tools/fuzz_features.py / tools/fuzz_features2.py (round 4: those sweeps found the start-of-batch starvation of crowd-capable
batches and the two-planner starvation).

run(seed)      kernel family (auto / role / wave / lat), overlapped launches, one or two launches per solve, lean mode, the LDS window,
               how n dependent solves are cut into calls (batches of any length, single solves, getters in between): the same n
               warm-started solves of B instances on a plain handle (one stream, automatic kernel, one call per solve) and on a randomly
               knobbed one -- costs, weights, U*, X*, the mean and the trajectory batch bit for bit.  Geometry, horizon (incl. the slow
               path), resolution (incl. the validated quotient), noise source and arithmetic vary with the seed, the same on both sides.
episode(seed)  the device-side closed loop on a knobbed handle against one stream / automatic kernel: the whole log
sampled(seed)  slip sampled per lookup (library draws): kernel variant, window, launches per solve, overlap
pair(seed)     TWO planners alive at once with their calls interleaved, each against its own plain run"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import numpy as np, torch
from benchnav_amd import NativeMPPI, _capi
from benchnav_amd.mppi import _DevArray


def case(seed):
    rng = np.random.default_rng(50_000 + seed)
    B = int(rng.choice([1, 1, 2, 3, 5, 16, 70]))
    K = int(rng.choice([64, 100, 257, 1000, 1024, 2049, 4160, 5000, 8192] if B <= 5 else [64, 100, 257, 1024]))
    T = int(rng.choice([1, 2, 5, 17, 33, 50, 64, 97, 150, 420]))
    if K * T * B > 3_000_000:
        T = max(1, 3_000_000 // (K * B))
    G = int(rng.choice([17, 50, 64, 129, 256]))
    res = float(rng.choice([0.5, 0.5, 0.25, 0.3, 0.1, 1.0]))
    x0, y0 = float(rng.choice([0.0, 0.0, -3.5, 1.3])), float(rng.choice([0.0, 0.0, 2.25]))
    wide = rng.random() < 0.07
    span = G * res * (1.25 if wide else 1.0)
    shared = bool(B > 1 and rng.random() < 0.4)
    maps = (rng.random((1 if shared else B, G, G)) * rng.choice([0.5, 0.95, 1.2])).astype(np.float32)
    states = np.stack([rng.uniform(x0, x0 + G * res, B), rng.uniform(y0, y0 + G * res, B), rng.uniform(-4, 4, B)], 1).astype(np.float32)
    goals = np.stack([rng.uniform(x0, x0 + G * res, B), rng.uniform(y0, y0 + G * res, B)], 1).astype(np.float32)
    noise = str(rng.choice(["philox", "philox", "kt2", "t2k"]))
    if noise != "philox" and B * K * T > 600_000:
        noise = "philox"
    n = int(rng.choice([1, 2, 3, 4, 7, 16, 17, 20, 33]))
    common = dict(horizon=T, num_samples=K, grid_size=G, resolution=res, x_limits=[x0, x0 + span], y_limits=[y0, y0 + span],
                  sigmas=[float(rng.choice([0.5, 0.1, 1.5])), float(rng.choice([0.5, 0.25, 2.0]))], lambda_=float(rng.choice([0.5, 0.05, 3.0])),
                  u_min=[float(rng.choice([0.0, -0.5])), float(rng.choice([-1.0, -0.3]))], u_max=[float(rng.choice([1.0, 0.4])), float(rng.choice([1.0, 2.0, 6.0]))],
                  dt=float(rng.choice([0.1, 0.1, 0.05, 0.4])), stuck_threshold=float(rng.choice([0.3, 0.0, 0.55])), num_instances=B, shared_map=shared,
                  seed=int(rng.integers(1, 1 << 30)), reference_order=bool(rng.random() < 0.25))
    knobs = dict(kernel=str(rng.choice(["auto", "auto", "role", "wave", "lat"])), overlap=bool(rng.random() < 0.7), pipeline=bool(rng.random() < 0.8),
                 lean=bool(rng.random() < 0.3), lds_window=bool(rng.random() < 0.85), store_controls=bool(rng.random() < 0.3))
    cuts = []                                               # how the n solves are cut into calls
    left = n
    while left:
        m = int(min(left, rng.choice([1, 1, 2, 3, 5, 16, 20, 33])))
        cuts.append((m, str(rng.choice(["none", "none", "weights", "sync", "flush", "first_action"]))))
        left -= m
    return dict(B=B, K=K, T=T, G=G, maps=maps, states=states, goals=goals, noise=noise, n=n, common=common, knobs=knobs, cuts=cuts, rng=rng)


def make(c, **kw):
    pl = NativeMPPI(**c["common"], **kw)
    for b in range(c["B"] if not c["common"]["shared_map"] else 1):
        pl.set_map(c["maps"][b], b if not c["common"]["shared_map"] else -1)
    for b in range(c["B"]):
        pl.set_goal(c["goals"][b], b)
    return pl


def outputs(pl, c, lean):
    pl.sync()
    B, T, K = c["B"], c["T"], c["K"]
    xs = torch.as_tensor(_DevArray(pl.device_buffer(_capi.BN_BUF_XSTAR)[0], (B, T + 1, 3)), device="cuda").cpu().numpy()
    us = torch.as_tensor(_DevArray(pl.device_buffer(_capi.BN_BUF_USTAR)[0], (B, T, 2)), device="cuda").cpu().numpy()
    out = {"xstar": xs, "ustar": us}
    for b in sorted({0, B // 2, B - 1}):
        out[f"cost{b}"] = pl.costs(b); out[f"w{b}"] = pl.weights(b); out[f"mean{b}"] = pl.get_mean(b)
        if not lean:
            out[f"X{b}"] = pl.states(b)
    return out


def run(seed):
    c = case(seed)
    B, K, T, n = c["B"], c["K"], c["T"], c["n"]
    st = torch.from_numpy(c["states"]).cuda()
    kind, eps, ring, stride = _capi.BN_NOISE_PHILOX, None, 1, 0
    if c["noise"] != "philox":
        ring = min(n, 3)
        shape = (ring, B, K, T, 2) if c["noise"] == "kt2" else (ring, B, T, 2, K)
        eps = torch.from_numpy(c["rng"].standard_normal(shape).astype(np.float32)).cuda()
        kind = _capi.BN_NOISE_DEVICE_KT2 if c["noise"] == "kt2" else _capi.BN_NOISE_DEVICE_T2K
        stride = eps[0].numel()
    torch.cuda.synchronize()
    eptr = eps.data_ptr() if eps is not None else None
    try:
        plain = make(c, overlap=False)
    except Exception as e:                                      # noqa: BLE001
        return "skip: " + str(e)[:80]
    with plain:
        for i in range(n):
            e_i = None if eps is None else eptr + 4 * stride * (i % ring)
            plain.solve_async_device(st.data_ptr(), e_i, kind)
        want = outputs(plain, c, False)
    if os.environ.get("FUZZ_BREAK"):                           # self-test of the sweep: a different Philox seed / goal must show up as mismatches
        c["common"] = dict(c["common"], seed=c["common"]["seed"] + 1); c["goals"] = c["goals"] + 0.25
    try:
        knobbed = make(c, **c["knobs"])
    except Exception as e:                                      # noqa: BLE001 -- a forced kernel that this geometry cannot run
        return "skip: " + str(e)[:80]
    with knobbed as pl:
        done = 0
        for m, then in c["cuts"]:
            # the ring position follows the solve count: batches start where the previous call stopped
            if eps is None:
                pl.solve_n_async_device(m, st.data_ptr()) if m > 1 else pl.solve_async_device(st.data_ptr())
            else:
                for j in range(m) if (done % ring) else [None]:
                    if j is None:
                        pl.solve_n_async_device(m, st.data_ptr(), eptr, kind, ring, stride)
                    else:
                        pl.solve_async_device(st.data_ptr(), eptr + 4 * stride * ((done + j) % ring), kind)
            done += m
            if then == "weights": pl.weights(0)
            elif then == "sync": pl.sync()
            elif then == "flush": pl.flush()
            elif then == "first_action": pl.first_action(B - 1)
        fa = np.stack([pl.first_action(b) for b in range(B)])      # the mail the tail posts: U*[0] of the latest solve, per instance
        got = outputs(pl, c, c["knobs"]["lean"])
        rec = pl.recovery_count()
        if not np.array_equal(fa, got["ustar"][:, 0, :], equal_nan=True):
            return "MISMATCH first_action vs ustar[:, 0]"
    for k, v in got.items():
        if not np.array_equal(v, want[k], equal_nan=True):
            d = np.abs(v.astype(np.float64) - want[k]).max() if np.isfinite(v).all() and np.isfinite(want[k]).all() else float("nan")
            return f"MISMATCH {k} max|d|={d:.3g}"
    return "ok" + (f" (recoveries {rec})" if rec else "")


def eq(a, b):
    return all(np.array_equal(x, y, equal_nan=True) for x, y in zip(a, b))


def episode(seed):
    c = case(seed)
    rng = np.random.default_rng(90_000 + seed)
    if c["K"] > 4096: c["common"]["num_samples"] = c["K"] = 1024
    if c["T"] > 100: c["common"]["horizon"] = c["T"] = 50
    c["common"]["reference_order"] = False if c["common"]["u_max"][1] * c["common"]["dt"] > 0.5 else c["common"]["reference_order"]
    if c["common"]["u_max"][1] * c["common"]["dt"] > 0.5: c["common"]["u_max"] = [c["common"]["u_max"][0], 1.0]; c["common"]["dt"] = 0.1
    steps = int(rng.choice([1, 2, 3, 5, 12, 20, 33]))
    n_maps = 1 if c["common"]["shared_map"] else c["B"]
    lm = np.clip(c["maps"], 0, 1).astype(np.float32)
    ls = (rng.random((n_maps, c["G"], c["G"])) * 0.1).astype(np.float32)
    thr, freeze = float(rng.choice([1.0, 0.3, 5.0, 20.0])), bool(rng.random() < 0.5)
    zdev = torch.from_numpy(rng.standard_normal((steps, c["B"])).astype(np.float32)).cuda() if rng.random() < 0.5 else None   # injected slip draws of the environment
    torch.cuda.synchronize()
    logs = []
    for knobs in (dict(overlap=False), dict(kernel=c["knobs"]["kernel"], overlap=True, lds_window=c["knobs"]["lds_window"])):
        try:
            pl = make(c, **knobs)
        except Exception as e:                                  # noqa: BLE001
            return "skip: " + str(e)[:60]
        with pl:
            try:
                pl.env_attach(lm, ls, goal_threshold=thr, delta_t=c["common"]["dt"], seed=7, freeze_on_goal=freeze)
                pl.episode(steps, c["states"], z_device_ptr=(zdev.data_ptr() if zdev is not None else None), wait=False)
                logs.append(pl.episode_log())
            except Exception as e:                              # noqa: BLE001
                return "skip: " + str(e)[:60]
    return "ok" if eq(logs[0], logs[1]) else "MISMATCH episode log"


def sampled(seed):
    c = case(seed)
    rng = np.random.default_rng(91_000 + seed)
    c["common"]["reference_order"] = False
    if c["T"] > 100: c["common"]["horizon"] = c["T"] = 64
    if c["B"] > 16: return "skip: big"
    sg = (rng.random((1 if c["common"]["shared_map"] else c["B"], c["G"], c["G"])) * rng.choice([0.0, 0.05, 0.3])).astype(np.float32)
    st = torch.from_numpy(c["states"]).cuda(); torch.cuda.synchronize()
    outs = []
    for knobs in (dict(overlap=False), dict(overlap=c["knobs"]["overlap"], pipeline=c["knobs"]["pipeline"], lds_window=c["knobs"]["lds_window"],
                                            store_controls=c["knobs"]["store_controls"])):
        try:
            pl = make(c, sampled_slip=True, **knobs)
        except Exception as e:                                  # noqa: BLE001
            return "skip: " + str(e)[:60]
        with pl:
            for b in range(sg.shape[0]):
                pl.set_slip_std(sg[b], b if not c["common"]["shared_map"] else -1)
            if len(outs) == 0:
                for _ in range(c["n"]): pl.solve_async_device(st.data_ptr())
            else:
                for m, then in c["cuts"]:
                    pl.solve_n_async_device(m, st.data_ptr()) if m > 1 else pl.solve_async_device(st.data_ptr())
                    if then == "weights": pl.weights(0)
                    elif then == "sync": pl.sync()
            outs.append(outputs(pl, c, False))
    for k in outs[0]:
        if not np.array_equal(outs[0][k], outs[1][k], equal_nan=True):
            return f"MISMATCH sampled {k}"
    return "ok"


def pair(seed):
    cs = [case(2 * seed + 100_000), case(2 * seed + 100_001)]
    for c in cs:
        if c["noise"] != "philox": c["noise"] = "philox"
    sts = [torch.from_numpy(c["states"]).cuda() for c in cs]
    torch.cuda.synchronize()
    want = []
    for c, st in zip(cs, sts):
        try:
            pl = make(c, overlap=False)
        except Exception as e:                                  # noqa: BLE001
            return "skip: " + str(e)[:60]
        with pl:
            for _ in range(c["n"]): pl.solve_async_device(st.data_ptr())
            want.append(outputs(pl, c, False))
    pls = []
    try:
        for c in cs: pls.append(make(c, **c["knobs"]))
    except Exception as e:                                      # noqa: BLE001
        for pl in pls: pl.close()
        return "skip: " + str(e)[:60]
    try:
        its = [iter(c["cuts"]) for c in cs]
        live = [True, True]
        while any(live):
            for j in (0, 1):
                if not live[j]: continue
                nxt = next(its[j], None)
                if nxt is None: live[j] = False; continue
                m, then = nxt
                pls[j].solve_n_async_device(m, sts[j].data_ptr()) if m > 1 else pls[j].solve_async_device(sts[j].data_ptr())
                if then == "weights": pls[j].weights(0)
                elif then == "sync": pls[j].sync()
                elif then == "flush": pls[j].flush()
                elif then == "first_action": pls[j].first_action(cs[j]["B"] - 1)
        rec = 0
        for j in (0, 1):
            got = outputs(pls[j], cs[j], cs[j]["knobs"]["lean"])
            rec += pls[j].recovery_count()
            for k, v in got.items():
                if not np.array_equal(v, want[j][k], equal_nan=True):
                    return f"MISMATCH pair[{j}] {k}"
    finally:
        for pl in pls: pl.close()
    return "ok" + (f" (recoveries {rec})" if rec else "")




def ops(seed):
    """Operation sequences: batches and single solves with setters (goal, mean, map), getters (top samples, controls) and -- on the
    knobbed side only -- the expired-wait test hook at random points: the journal's re-run has to reproduce, through whatever came in
    between, what a handle that never overlapped computes.  Philox noise; lean handles re-roll their top samples."""
    c = case(seed + 300_000)
    rng = np.random.default_rng(93_000 + seed)
    c["noise"] = "philox"
    B, K, T, G = c["B"], c["K"], c["T"], c["G"]
    if B > 16:
        return "skip: big"
    st_all = [c["states"]] + [(c["states"] + rng.normal(0, 0.3, c["states"].shape)).astype(np.float32) for _ in range(2)]
    goals2 = (c["goals"] + rng.normal(0, 2.0, c["goals"].shape)).astype(np.float32)
    mean2 = (rng.standard_normal((T, 2)) * 0.3).astype(np.float32)
    map2 = np.clip(c["maps"][0] * 0.5 + 0.2, 0, 1).astype(np.float32)
    script = []
    for _ in range(int(rng.integers(2, 9))):
        kind = str(rng.choice(["batch", "batch", "batch", "single", "goal", "mean", "map", "top", "state", "expire", "expire", "sync"]))
        script.append((kind, int(rng.choice([2, 3, 4, 5, 7, 16, 20])), int(rng.integers(0, B)), int(rng.integers(0, 3))))
    script.append(("batch", int(rng.choice([3, 5, 17])), 0, 0))
    tops = [[], []]
    outs = []
    cur = torch.cuda.current_stream().cuda_stream            # the handles run on torch's stream: the 'state' copies below are stream-ordered
    for side, knobs in enumerate((dict(overlap=False), c["knobs"])):
        try:
            pl = make(c, stream=cur, **knobs)
        except Exception as e:                                  # noqa: BLE001
            return "skip: " + str(e)[:60]
        with pl:
            st = torch.from_numpy(st_all[0].copy()).cuda(); torch.cuda.synchronize()
            solves = 0
            last = ""
            fresh_map = False
            for kind, m, b, which in script:
                prev, last = last, kind
                if kind == "batch":
                    if side == 0:
                        for _ in range(m): pl.solve_async_device(st.data_ptr())
                    else:
                        pl.solve_n_async_device(m, st.data_ptr())
                    solves += m; fresh_map = False
                elif kind == "single":
                    pl.solve_async_device(st.data_ptr()); solves += 1; fresh_map = False
                elif kind == "goal":
                    pl.set_goal(goals2[b], b)
                elif kind == "mean":
                    pl.set_mean(mean2, b)
                elif kind == "map":
                    pl.set_map(map2, -1 if c["common"]["shared_map"] else b); fresh_map = True
                elif kind == "top" and solves and not fresh_map:    # (a lean handle refuses to re-roll on a map its solve did not see)
                    tops[side].append(pl.top_samples(min(5, K), b))
                elif kind == "state":
                    # the caller rewrites its state buffer in stream order (allowed: the journal keeps what each batch was given)
                    st.copy_(torch.from_numpy(st_all[which]).cuda(), non_blocking=True)
                elif kind == "expire" and side == 1 and prev == "batch":
                    # (the hook stands for an expiry in the batch just enqueued; with nothing journalled -- the knobbed handle did not
                    # overlap -- it leaves NaN patterns nothing can repair: not a case)
                    _capi.check(pl._lib.bn_mppi_debug_expire_wait(pl._h))
                    try:
                        pl.sync()
                    except Exception as e:                      # noqa: BLE001
                        if "cannot be re-run" in str(e):
                            return "skip: nothing journalled"
                        raise
                elif kind == "sync":
                    pl.sync()
            outs.append(outputs(pl, c, side == 1 and c["knobs"]["lean"]))
    diff = [k for k, v in outs[1].items() if not np.array_equal(v, outs[0][k], equal_nan=True)]
    if diff:
        return "MISMATCH ops " + ",".join(diff)
    if len(tops[0]) != len(tops[1]):
        return "MISMATCH ops top count"
    for (s0, w0), (s1, w1) in zip(*tops):
        if not (np.array_equal(s0, s1, equal_nan=True) and np.array_equal(w0, w1, equal_nan=True)):
            return "MISMATCH ops top samples"
    return "ok"
