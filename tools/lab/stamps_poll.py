"""How often does a waiting launch look?  Timing build: the cycle counter at the end of every granule poll of workgroup 0's three polling
waves (chain, wave 2, wave 3) in the overlapped steady state -- the last 16 polls before the rows were seen.  BN_VARIANT selects the library."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes as C, numpy as np, torch
from benchnav_amd import build as b
b.LIB_PATH = os.path.join(ROOT, "tools", "_ablate", "lib_%s.so" % os.environ.get("BN_VARIANT", "timing"))
from benchnav_amd import NativeMPPI, synth
inst = synth.make_instance(256, seed=0)
pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, kernel="lat")
pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
stamps = torch.zeros(1024, dtype=torch.int64, device="cuda")
pl._lib.bn_mppi_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
pl._lib.bn_mppi_debug_set_stamps(pl._h, C.c_void_p(stamps.data_ptr()))
st = inst.start.cuda(); torch.cuda.synchronize()
for rep in range(4):
    stamps.zero_(); pl.solve_n_async_device(201, st.data_ptr()); pl.sync()
    s = stamps.cpu().numpy()
    for w in range(3):
        ring = s[800 + 32 * w:800 + 32 * w + 16]; last = int(s[800 + 32 * w + 16])
        order = [ring[(last - k) & 15] for k in range(min(last + 1, 16))][::-1]
        d = np.diff(np.array(order, dtype=np.int64)) / 2400.0
        print(f"wave {w + 1}: {last + 1} polls; intervals of the last ones (us):", " ".join(f"{x:.2f}" for x in d))
