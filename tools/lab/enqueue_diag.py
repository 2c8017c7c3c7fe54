"""Is the sporadic 2x-slower timed loop host-bound?  Time the enqueue call (returns when all launches are queued)
against the whole loop, a few trials in one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from benchnav_amd import NativeMPPI, synth
inst = synth.make_instance(256, seed=0)
st = inst.start.cuda()
for trial in range(12):
    pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, stream=torch.cuda.current_stream().cuda_stream)
    pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
    pl.solve_n_async_device(200, st.data_ptr()); pl.flush(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    pl.solve_n_async_device(3000, st.data_ptr())
    t1 = time.perf_counter()
    pl.flush(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"trial {trial}: enqueue {1e6*(t1-t0)/3000:.2f} us/launch   total {1e6*(t2-t0)/3000:.2f} us/solve   load {os.getloadavg()[0]:.1f}")
    pl.close()
