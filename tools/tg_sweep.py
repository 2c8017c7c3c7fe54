"""Dependent solves of ONE instance over the horizon T and the map size G (K=1024): looking for cliffs between code paths."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from benchnav_amd import NativeMPPI, synth
torch.set_num_threads(1)
def rate(K, T, G, res=0.5, **kw):
    inst = synth.make_instance(G, seed=0, resolution=res)
    st = inst.start.cuda()
    with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=res, **kw) as pl:
        pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
        pl.solve_n_async_device(50, st.data_ptr()); pl.sync()
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            pl.solve_n_async_device(300, st.data_ptr()); pl.sync()
            best = min(best, (time.perf_counter() - t0) / 300)
    return best * 1e6
for T in (5, 10, 20, 35, 50, 64, 80, 100, 150, 200, 300):
    try:
        print(f"T={T:4d} K=1024 G=256: {rate(1024, T, 256):7.2f} us per solve ({rate(1024, T, 256) / T * 1e3:6.1f} ns per step)", flush=True)
    except Exception as e:
        print(f"T={T}: {str(e)[:100]}", flush=True)
for G in (32, 64, 128, 256, 512, 1024, 2048):
    print(f"G={G:5d} K=1024 T=50: {rate(1024, 50, G):7.2f} us per solve", flush=True)
for res in (0.5, 0.3, 0.25, 1.0):
    print(f"res={res} K=1024 T=50 G=256: {rate(1024, 50, 256, res=res):7.2f} us per solve", flush=True)
