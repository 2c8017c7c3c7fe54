"""Dependent solves per second in both arithmetics (default / BN_FLAG_REFERENCE_ORDER): one instance (latency kernel), 64 instances
(role kernel), 256 instances (one-wave kernel), the reference's K=5000 point and configs[4] (ticket paths)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from benchnav_amd import NativeMPPI, synth
torch.set_num_threads(1)
def rate(K, T, G, B, ref, n=300):
    inst = synth.make_instance(G, seed=0)
    st = torch.stack([inst.start] * B).cuda()
    with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, num_instances=B, shared_map=True, reference_order=ref) as pl:
        pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
        pl.solve_n_async_device(50, st.data_ptr()); pl.sync()
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            pl.solve_n_async_device(n, st.data_ptr()); pl.sync()
            best = min(best, (time.perf_counter() - t0) / n)
        return best * 1e6, pl.launches_per_solve()
for K, T, G, B in ((1024, 50, 256, 1), (1024, 50, 256, 64), (1024, 50, 256, 256), (5000, 50, 64, 1), (16384, 100, 512, 1)):
    a, _ = rate(K, T, G, B, False)
    b, l = rate(K, T, G, B, True)
    print(f"K={K} T={T} G={G} B={B}: default {a:7.2f} us per launch, reference order {b:7.2f} us ({b / a:.2f} x, {l} launch(es) per solve)", flush=True)
