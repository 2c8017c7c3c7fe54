import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from benchnav_amd import NativeMPPI, _capi, synth
G,K,T,RES=256,1024,50,0.5
insts=[synth.make_instance(G, seed=s, resolution=RES, jitter=True) for s in range(4)]
def run(B, pipeline, noise, shared, n=40):
    pl = NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=RES, num_instances=B, stream=0, shared_map=shared, pipeline=pipeline, profile=True)
    if shared: pl.set_map(insts[0].risk.numpy()); pl.set_goal(insts[0].goal.numpy())
    else:
        for b in range(B): pl.set_map(insts[b%4].risk.numpy(), b); pl.set_goal(insts[b%4].goal.numpy(), b)
    states=torch.stack([insts[b%4].start for b in range(B)]).cuda()
    eps=torch.randn(B,T,2,K,device="cuda"); torch.cuda.synchronize()
    out=[]
    for r in range(3):
        t0=time.perf_counter()
        for i in range(n):
            if noise=="philox": pl.solve_async_device(states.data_ptr())
            else: pl.solve_async_device(states.data_ptr(), eps.data_ptr(), _capi.BN_NOISE_DEVICE_T2K)
        pl.sync(); out.append((time.perf_counter()-t0)/n*1e6)
    print(f"B={B} pipeline={pipeline} noise={noise} shared={shared}: us/launch", " ".join(f"{o:.0f}" for o in out), "kernel_ms", pl.kernel_ms(), flush=True)
    pl.close()
for B in (1, 8, 64):
    for pipeline in (True, False):
        for noise in ("philox","t2k"):
            run(B, pipeline, noise, False)
run(64, True, "philox", True)
