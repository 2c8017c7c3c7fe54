import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from benchnav_amd import NativeMPPI, _capi, synth
G,K,T,RES=256,1024,50,0.5
insts=[synth.make_instance(G, seed=s, resolution=RES, jitter=True) for s in range(4)]
def run(B, stream, shared, distinct_states, reps=4, n=50, tag=""):
    pl = NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=RES, num_instances=B, stream=stream, shared_map=shared)
    if shared:
        pl.set_map(insts[0].risk.numpy()); pl.set_goal(insts[0].goal.numpy())
    else:
        for b in range(B):
            it=insts[b%4]; pl.set_map(it.risk.numpy(), b); pl.set_goal(it.goal.numpy(), b)
    states=torch.stack([insts[b%4 if distinct_states else 0].start for b in range(B)]).cuda()
    torch.cuda.synchronize()
    out=[]
    for r in range(reps):
        t0=time.perf_counter()
        for i in range(n): pl.solve_async_device(states.data_ptr())
        pl.sync(); out.append((time.perf_counter()-t0)/n*1e6)
    print(f"{tag} B={B} stream={stream} shared={shared} distinct_states={distinct_states}: us/solve:", " ".join(f"{o:.0f}" for o in out), flush=True)
    pl.close()
for B in (8, 64):
    for shared in (True, False):
        for ds in (False, True):
            run(B, None, shared, ds); run(B, 0, shared, ds)
