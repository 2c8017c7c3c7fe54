import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from benchnav_amd import NativeMPPI, synth
B, K, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
inst = synth.make_instance(256, seed=0)
pl = NativeMPPI(horizon=T, num_samples=K, grid_size=256, resolution=0.5, num_instances=B, shared_map=True)
pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
st = torch.stack([inst.start] * B).cuda(); torch.cuda.synchronize()
pl.solve_n_async_device(16, st.data_ptr()); pl.solve_async_device(st.data_ptr()); pl.solve_n_async_device(5, st.data_ptr())
pl.sync()
print(f"B={B} K={K} T={T}: recoveries {pl.recovery_count()}", flush=True)
pl.close()
