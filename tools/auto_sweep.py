"""Is the automatic kernel choice the best one?  For a grid of (K, T, B): microseconds per launch of dependent solves with the
automatic choice (overlapped / one stream) and with each kernel forced (overlapped); flags cells where `auto` is >5 % off the best."""
import os, sys, time, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from benchnav_amd import NativeMPPI, synth
torch.set_num_threads(1)
inst = synth.make_instance(256, seed=0, resolution=0.5)
def rate(K, T, B, **kw):
    try:
        with NativeMPPI(horizon=T, num_samples=K, grid_size=256, resolution=0.5, num_instances=B, shared_map=True, lean=bool(int(os.environ.get("BN_LEAN", "1"))), **kw) as pl:
            pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
            st = torch.stack([inst.start] * B).cuda()
            n = max(20, min(300, int(3e4 / (B * K / 1024 * T / 50 + 10))))
            pl.solve_n_async_device(20, st.data_ptr()); pl.sync()
            best = 1e9
            for _ in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                pl.solve_n_async_device(n, st.data_ptr()); pl.sync()
                best = min(best, (time.perf_counter() - t0) / n)
        return best * 1e6
    except Exception as e:
        return float("nan")
Ks = [int(x) for x in os.environ.get("BN_KS", "128,512,1024,2048,4096").split(",")]
Ts = [int(x) for x in os.environ.get("BN_TS", "10,20,50,100").split(",")]
Bs = [int(x) for x in os.environ.get("BN_BS", "1,4,16,48,128,300").split(",")]
for K, T, B in itertools.product(Ks, Ts, Bs):
    if B * K * T > 300 * 4096 * 50: continue
    r = {"auto": rate(K, T, B), "auto-1s": rate(K, T, B, overlap=False)}
    for kern in ("lat", "role", "wave"):
        r[kern] = rate(K, T, B, kernel=kern)
    vals = {k: v for k, v in r.items() if v == v}
    best = min(vals, key=vals.get)
    flag = "" if r["auto"] <= 1.05 * vals[best] else f"   <-- {best} is {100 * (r['auto'] / vals[best] - 1):.0f} % faster"
    print(f"K={K:5d} T={T:3d} B={B:3d}: " + "  ".join(f"{k} {v:7.2f}" for k, v in r.items()) + flag, flush=True)
