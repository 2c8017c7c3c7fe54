"""Wall-clock rate of dependent solves enqueued in one call, overlapped launches on / off (K=1024, T=50, 256x256)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from benchnav_amd import NativeMPPI, synth
torch.set_num_threads(1)
inst = synth.make_instance(256, seed=0)
st = inst.start.cuda()
for lean in (False, True):
    for overlap in (False, True):
        pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, stream=0, overlap=overlap, lean=lean)
        pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
        pl.solve_n_async_device(300, st.data_ptr()); pl.sync()
        best = 1e9
        for n in (3000, 3000, 3000, 20, 20, 20):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            pl.solve_n_async_device(n, st.data_ptr()); pl.flush(); torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            print(f"lean={lean!s:5s} overlap={overlap!s:5s} n={n:5d}: {dt * 1e6:6.2f} us per solve  ({1 / dt:8.0f} solves/s)", flush=True)
        pl.close()
