"""Dependent solves of ONE instance over the number of rollouts K (T=50, 256x256): which path each K takes and what it costs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from benchnav_amd import NativeMPPI, synth
torch.set_num_threads(1)
inst = synth.make_instance(256, seed=0, resolution=0.5)
st = inst.start.cuda()
for K in [int(x) for x in os.environ.get("BN_KS", "512,1024,2048,3072,4096,8192,16384,32768").split(",")]:
    row = []
    for overlap in (True, False):
        with NativeMPPI(horizon=50, num_samples=K, grid_size=256, resolution=0.5, overlap=overlap) as pl:
            pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
            pl.solve_n_async_device(50, st.data_ptr()); pl.sync()
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                pl.solve_n_async_device(300, st.data_ptr()); pl.sync()
                best = min(best, (time.perf_counter() - t0) / 300)
            row.append(best * 1e6)
    print(f"K={K:6d}: {row[0]:6.2f} us per solve (one stream {row[1]:6.2f})  {K / row[0]:7.1f} rollouts/us", flush=True)
