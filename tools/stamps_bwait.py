import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C, numpy as np, torch
from benchnav_amd import build as b
b.LIB_PATH = os.path.join(ROOT, "tools", "_ablate", "lib_%s.so" % os.environ.get("BN_VARIANT", "t_bwait"))
from benchnav_amd import NativeMPPI, synth
inst = synth.make_instance(256, seed=0)
pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, kernel="lat")
pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
stamps = torch.zeros(1024, dtype=torch.int64, device="cuda")
pl._lib.bn_mppi_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
pl._lib.bn_mppi_debug_set_stamps(pl._h, C.c_void_p(stamps.data_ptr()))
st = inst.start.cuda(); torch.cuda.synchronize()
for rep in range(3):
    stamps.zero_(); pl.solve_n_async_device(201, st.data_ptr()); pl.sync()
    s = stamps.cpu().numpy()
    w = s[192:192 + 60].reshape(5, 12)
    c0 = w[1, 9]
    print(f"chain last step {(w[1, 11] - c0) / 2400:.2f} | B start {(w[3, 7] - c0) / 2400:.2f} | B out {(w[3, 0] - c0) / 2400:.2f} -> {(w[3, 0] - w[3, 7]) / 2.4 / 51:.1f} ns a slot")
