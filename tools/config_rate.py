"""Dependent solves per second of the big single-instance configurations (K=16384 T=100 G=512; sampled slip K=8192) and the
64-instance batch, for build variants (tools/build_variant.py):  python tools/config_rate.py <variant|main>
(ONE library per process: two images of the library in one process contend for the per-device lock and the second one measures slow)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from benchnav_amd import build as b
from benchnav_amd import _capi
torch.set_num_threads(1)
MAIN = b.LIB_PATH
def rate(pl, st, n):
    pl.solve_n_async_device(30, st.data_ptr()); pl.sync()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pl.solve_n_async_device(n, st.data_ptr()); pl.sync()
        best = min(best, (time.perf_counter() - t0) / n)
    return best * 1e6
for name in sys.argv[1:]:
    b.LIB_PATH = MAIN if name == "main" else os.path.join(ROOT, "tools", "_ablate", f"lib_{name}.so")
    _capi._lib = None
    from benchnav_amd import NativeMPPI, synth
    out = []
    inst = synth.make_instance(512, seed=0, resolution=0.5)
    with NativeMPPI(horizon=100, num_samples=16384, grid_size=512, resolution=0.5) as pl:
        pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
        out.append(f"c5 {rate(pl, inst.start.cuda(), 300):6.2f} us")
    inst = synth.make_instance(256, seed=0, resolution=0.5)
    with NativeMPPI(horizon=50, num_samples=8192, grid_size=256, resolution=0.5, sampled_slip=True) as pl:
        pl.set_map(inst.risk.numpy()); pl.set_slip_std(synth.slip_std_map(256, seed=0).numpy()); pl.set_goal(inst.goal.numpy())
        out.append(f"sampled {rate(pl, inst.start.cuda(), 300):6.2f} us")
    with NativeMPPI(horizon=50, num_samples=4096, grid_size=256, resolution=0.5) as pl:
        pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
        out.append(f"K=4096 {rate(pl, inst.start.cuda(), 300):6.2f} us")
    B = 64
    with NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, num_instances=B, shared_map=True) as pl:
        pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
        out.append(f"B=64 {rate(pl, torch.stack([inst.start] * B).cuda(), 300):6.2f} us")
    for B, lean in ((64, True), (256, False), (256, True)):
        with NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, num_instances=B, shared_map=True, lean=lean) as pl:
            pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
            out.append(f"B={B}{' lean' if lean else ''} {rate(pl, torch.stack([inst.start] * B).cuda(), 300):6.2f} us")
    print(f"{name:10s} " + " | ".join(out), flush=True)
