"""Two consecutive overlapped launches on the chip-wide wall clock (timing build): entry, first and last step of the chain, the barrier
behind the cost, partial rows out -- of workgroup 0 of solve n-2 (same stream as n), n-1 and n, relative to the entry of solve n-1.
Answers: how long before its predecessor's rows is a launch resident, and what does its period hang on.  BN_VARIANT selects the library."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes as C, numpy as np, torch
from benchnav_amd import build as b
b.LIB_PATH = os.path.join(ROOT, "tools", "_ablate", "lib_%s.so" % os.environ.get("BN_VARIANT", "timing"))
from benchnav_amd import NativeMPPI, synth
inst = synth.make_instance(256, seed=0)
pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, kernel="lat", reference_order=bool(int(os.environ.get("BN_REF", "0"))))
pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
stamps = torch.zeros(1024, dtype=torch.int64, device="cuda")
pl._lib.bn_mppi_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
pl._lib.bn_mppi_debug_set_stamps(pl._h, C.c_void_p(stamps.data_ptr()))
st = inst.start.cuda(); torch.cuda.synchronize()
rows = []
for rep in range(21):
    pl.solve_n_async_device(201, st.data_ptr()); pl.sync()
    s = stamps.cpu().numpy().astype(np.float64)
    n = pl.solve_count()
    last, prev = s[32 + ((n - 1) & 1) * 16:][:16], s[32 + ((n - 2) & 1) * 16:][:16]
    e0 = prev[0]
    rows.append([(prev[i] - e0) / 100.0 for i in (0, 1, 2, 3, 4, 5)] + [(last[i] - e0) / 100.0 for i in (0, 1, 2, 3, 4, 5)])
r = np.median(np.stack(rows), axis=0)
names = ("entry", "prologue done (wave 0)", "chain: chunk 0 done", "chain: last step", "behind the e barrier", "rows out")
print("us after the entry of solve n-1:      solve n-1      solve n     (n - (n-1))")
for i, nm in enumerate(names):
    print(f"  {nm:26s} {r[i]:10.2f} {r[6 + i]:12.2f} {r[6 + i] - r[i]:12.2f}")
