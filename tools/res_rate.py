"""Dependent solves of one instance (K=1024, T=50, G=256) over the map resolution: power of two (exact multiply) against general
resolutions with the validated three-instruction quotient (the IEEE division it replaced: 13.1 us at res 0.3)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from benchnav_amd import build as _b
if os.environ.get("BN_TOOL_LIB", "main") != "main":                     # a variant built by tools/build_variant_fast.py <name>
    _b.LIB_PATH = os.path.join(ROOT, "tools", "_ablate", "lib_%s.so" % os.environ["BN_TOOL_LIB"])
from benchnav_amd import NativeMPPI, synth
torch.set_num_threads(1)
def rate(K, T, G, res, B=1, **kw):
    inst = synth.make_instance(G, seed=0, resolution=res)
    st = torch.stack([inst.start] * B).cuda()
    with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=res, num_instances=B, shared_map=True, **kw) as pl:
        fq = pl.fast_quotient()
        pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
        pl.solve_n_async_device(50, st.data_ptr()); pl.sync()
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            pl.solve_n_async_device(300, st.data_ptr()); pl.sync()
            best = min(best, (time.perf_counter() - t0) / 300)
    return best * 1e6, fq
for B in (1, 64):
    for res in (0.5, 0.3, 0.7):
        us, fq = rate(1024, 50, 256, res, B)
        print(f"B={B:3d} res={res}: {us:7.2f} us per launch (quotient mode {fq})", flush=True)
