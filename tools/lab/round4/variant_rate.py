"""Overlapped dependent-solve period of build variants (tools/build_variant.py):  python tools/variant_rate.py base skip1 skip2 ...
(K=1024, T=50, 256x256; best of three batches of 3000).  Variants built with -DBN_VAR_SKIP=... give wrong results on purpose:
the difference to `base` is what that piece costs on the critical path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from benchnav_amd import build as b
from benchnav_amd import _capi
torch.set_num_threads(1)
lean = bool(int(os.environ.get("BN_LEAN", "0")))
for name in sys.argv[1:]:
    b.LIB_PATH = os.path.join(ROOT, "tools", "_ablate", f"lib_{name}.so")
    _capi._lib = None
    from benchnav_amd import NativeMPPI, synth
    inst = synth.make_instance(256, seed=0)
    st = inst.start.cuda()
    pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, stream=0, lean=lean)
    pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
    pl.solve_n_async_device(300, st.data_ptr()); pl.sync()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pl.solve_n_async_device(3000, st.data_ptr()); pl.flush(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 3000)
    print(f"{name:12s} {best * 1e6:6.2f} us per solve", flush=True)
    pl.close()
