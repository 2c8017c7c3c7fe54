"""Exploratory GPU run: parity of every golden case + a first timing (not a test, not the bench)."""
import os, sys, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")]
import numpy as np
from helpers import CASES, load_case, oracle_params_for, parity_metrics
from oracle import oracle as O
from benchnav_amd import NativeMPPI

def planner_for(fx, **kw):
    return NativeMPPI(horizon=int(fx["T"]), num_samples=int(fx["K"]), grid_size=int(fx["G"]), resolution=float(fx["res"]),
                      x_limits=fx["x_limits"].tolist(), y_limits=fx["y_limits"].tolist(), sigmas=fx["sigmas"].tolist(),
                      inv_var=fx["inv_var"].tolist(), lambda_=float(fx["lam"]), u_min=fx["u_min"].tolist(), u_max=fx["u_max"].tolist(),
                      stuck_threshold=float(fx["thr"]), store_controls=True, **kw)

for name in CASES:
    fx = load_case(name)
    for lds in (True, False):
        pl = planner_for(fx, lds_window=lds)
        pl.set_map(fx["R"]); pl.set_goal(fx["goal"])
        p = oracle_params_for(fx, O.TRIG_SPEC)
        for i in range(int(fx["n_solves"])):
            pl.set_mean(fx[f"mean_{i}"])
            us, xs = pl.solve(fx[f"state_{i}"], fx[f"eps_{i}"])
            got = dict(U=pl.controls(), X=pl.states(), cost=pl.costs(), w=pl.weights(), Ustar=us[0], Xstar=xs[0])
            orc = O.solve(p, fx["R"], fx[f"state_{i}"], fx[f"mean_{i}"], fx[f"eps_{i}"])
            bit = {k: bool(np.array_equal(got[k], orc[k])) for k in ("U", "X", "cost")}
            dev = {k: float(np.abs(got[k] - orc[k]).max()) for k in ("cost", "w", "Ustar", "Xstar")}
            m = parity_metrics(got, fx, i)
            print(f"{name:9s} lds={int(lds)} solve {i} bitexact {bit} vs-oracle {dev} | vs-ref X {m['X_max']:.1e} out {m['cost_outliers']} w {m['w_max']:.1e} U* {m['Ustar_max']:.1e} X* {m['Xstar_max']:.1e}")
            nm = pl.get_mean()
            assert np.array_equal(nm, us[0]), "mean not updated to U*"
        pl.close()

# timing at config 2
fx = load_case("c2")
for label, eps in (("injected-host", fx["eps_0"]), ("philox", None)):
    pl = planner_for(fx); pl.set_map(fx["R"]); pl.set_goal(fx["goal"])
    for _ in range(5): pl.solve(fx["state_0"], eps)
    t = time.perf_counter(); n = 50
    for _ in range(n): pl.solve(fx["state_0"], eps)
    dt = (time.perf_counter() - t) / n
    print(f"sync solve {label}: {dt*1e6:.1f} us/solve")
    pl.close()
pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, profile=True)
pl.set_map(fx["R"]); pl.set_goal(fx["goal"])
import torch
st = torch.tensor(fx["state_0"], device="cuda")
for _ in range(20): pl.solve_async_device(st.data_ptr())
pl.sync(); pl.kernel_ms()
t = time.perf_counter(); n = 500
for _ in range(n): pl.solve_async_device(st.data_ptr())
pl.sync(); dt = (time.perf_counter() - t) / n
print(f"async philox profiled: {dt*1e6:.1f} us/solve; kernel ms (rollout, finish, n) = {pl.kernel_ms()}")
pl.close()
pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5)
pl.set_map(fx["R"]); pl.set_goal(fx["goal"])
for _ in range(20): pl.solve_async_device(st.data_ptr())
pl.sync()
t = time.perf_counter(); n = 2000
for _ in range(n): pl.solve_async_device(st.data_ptr())
pl.sync(); dt = (time.perf_counter() - t) / n
print(f"async philox unprofiled: {dt*1e6:.1f} us/solve -> {1/dt:.0f} solves/s")
