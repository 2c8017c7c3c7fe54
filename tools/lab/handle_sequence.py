"""Eighteen 64-instance handles created, run and destroyed one after another on torch's current (null) stream: warm-up batch of NW solves,
three timed batches of NB.  Prints the per-launch time and bn_mppi_recovery_count() of every handle.  With even NW / NB the first solve
of a batch used to land on the handle's internal stream, whose queue wakes up later than the handle's own: solve 1 overtook solve 0, its
waiting workgroups crowded solve 0 out and the bounded waits expired -- 3-7 recoveries per 18 handles (round 3 and early round 4), none
with odd lengths.  Since the fix (big launches always start on the handle's own stream, mppi_capi.cpp) none either way.
    NW=30 NB=400 python tools/handle_sequence.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from benchnav_amd import NativeMPPI, synth
torch.set_num_threads(1)
G,K,T,B=256,1024,50,64
NW=int(os.environ.get("NW","30")); NB=int(os.environ.get("NB","400"))
insts=[synth.make_instance(G, seed=s, resolution=0.5, jitter=True) for s in range(B)]
states=torch.stack([it.start for it in insts]).cuda()
stream=torch.cuda.current_stream()
def run(st, lean=False, overlap=True, **kw):
    pl=NativeMPPI(horizon=T,num_samples=K,grid_size=G,resolution=0.5,num_instances=B, stream=st, lean=lean, overlap=overlap, **kw)
    for b,it in enumerate(insts):
        pl.set_map(it.risk.numpy(), b); pl.set_goal(it.goal.numpy(), b)
    pl.solve_n_async_device(NW, states.data_ptr()); pl.flush(); torch.cuda.synchronize(); pl.sync()
    best=1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0=time.perf_counter()
        pl.solve_n_async_device(NB, states.data_ptr()); pl.flush(); torch.cuda.synchronize()
        best=min(best,(time.perf_counter()-t0)/NB); pl.sync()
    print('stream',st,'lean',lean,'overlap',overlap,kw,'%.2f us'%(best*1e6),'recov',pl.recovery_count(), flush=True)
    pl.close()
for rep in range(3):
    for st in (stream.cuda_stream,):
        run(st); run(st, lean=True); run(st, overlap=False); run(st); run(st); run(st)
