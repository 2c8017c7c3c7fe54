"""Dependent solves of B instances per launch (K=1024, T=50, 256x256), enqueued in one call: overlapped launches on / off."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from benchnav_amd import NativeMPPI, synth
torch.set_num_threads(1)
for B in [int(x) for x in os.environ.get("BN_BS", "16,32,64,128").split(",")]:
    insts = [synth.make_instance(256, seed=b) for b in range(B)]
    st = torch.stack([it.start for it in insts]).cuda()
    for lean in (False, True):
        for overlap in (False, True):
            pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, num_instances=B, overlap=overlap, lean=lean,
                            kernel=os.environ.get("BN_KERNEL", "auto"))
            for b, it in enumerate(insts):
                pl.set_map(it.risk.numpy(), b); pl.set_goal(it.goal.numpy(), b)
            pl.solve_n_async_device(50, st.data_ptr()); pl.sync()
            best, worst = 1e9, 0.0
            for _ in range(int(os.environ.get("BN_REPS", "3"))):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                pl.solve_n_async_device(400, st.data_ptr()); pl.sync()
                dt_ = (time.perf_counter() - t0) / 400
                best, worst = min(best, dt_), max(worst, dt_)
            by = pl.algorithmic_bytes(injected_noise=False) * B
            print(f"B={B:4d} lean={lean!s:5s} overlap={overlap!s:5s}: {best * 1e6:7.2f} us per launch  {B / best / 1e6:5.2f} M solves/s  {by / best / 1e12:5.2f} TB/s   (slowest batch {worst * 1e6:7.2f} us)", flush=True)
            pl.close()
