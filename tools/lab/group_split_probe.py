"""Would splitting a many-instance launch into instance groups on separate streams raise throughput?  G independent handles of
64/G instances each (private streams), n dependent solves enqueued on each, all running at once; against one 64-instance handle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from benchnav_amd import NativeMPPI, synth
torch.set_num_threads(1)
Btot = int(os.environ.get("BN_B", "64"))
lean = bool(int(os.environ.get("BN_LEAN", "0")))
kern = os.environ.get("BN_KERNEL", "auto")
for G in (1, 2, 4, 8):
    nb = Btot // G
    hs = []
    for g in range(G):
        insts = [synth.make_instance(256, seed=g * nb + b) for b in range(nb)]
        pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, num_instances=nb, lean=lean, kernel=kern)
        for b, it in enumerate(insts):
            pl.set_map(it.risk.numpy(), b); pl.set_goal(it.goal.numpy(), b)
        st = torch.stack([it.start for it in insts]).cuda()
        hs.append((pl, st))
    torch.cuda.synchronize()
    for pl, st in hs: pl.solve_n_async_device(50, st.data_ptr())
    for pl, st in hs: pl.sync()
    best = 1e9
    n = 300
    for _ in range(3):
        t0 = time.perf_counter()
        for pl, st in hs: pl.solve_n_async_device(n, st.data_ptr())
        for pl, st in hs: pl.sync()
        best = min(best, (time.perf_counter() - t0) / n)
    print(f"{G} group(s) of {nb:3d}: {best * 1e6:6.2f} us per {Btot} solves  ({Btot / best / 1e6:.2f} M solves/s)", flush=True)
    for pl, st in hs: pl.close()
