"""Per-workgroup timeline of THREE consecutive launches of an overlapped batch of the latency kernel (timing build: tools/build_variant_fast.py
--timing timing): entry / exit of every workgroup on the chip-wide clock (the 16 rollout workgroups and the aux workgroup that runs the
previous solve's tail).  What a launch's end waits for, and how long after it the next launch of the same stream enters."""
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
nb = 32
stamps = torch.zeros(64 + 4 * 2 * nb + 64, dtype=torch.int64, device="cuda")
pl._lib.bn_mppi_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
pl._lib.bn_mppi_debug_trace_by_parity.argtypes = [C.c_void_p, C.c_int]
pl._lib.bn_mppi_debug_set_stamps(pl._h, C.c_void_p(stamps.data_ptr()))
pl._lib.bn_mppi_debug_trace_by_parity(pl._h, 1)
st = inst.start.cuda(); torch.cuda.synchronize()
rows = []
for rep in range(15):
    stamps.zero_()
    pl.solve_n_async_device(101, st.data_ptr()); pl.sync()
    total = pl.solve_count()
    r = stamps.cpu().numpy()[64:64 + 4 * 2 * nb].reshape(2, nb, 4).astype(np.float64)
    last, prev = r[(total - 1) & 1], r[(total - 2) & 1]
    base = prev[:16, 0][prev[:16, 0] > 0].min()
    f = lambda rr: [(rr[:16, 0][rr[:16, 0] > 0].min() - base) / 100, (rr[:16, 0].max() - base) / 100, (rr[:16, 1][rr[:16, 1] > 0].min() - base) / 100, (rr[:16, 1].max() - base) / 100,
                    (rr[16, 0] - base) / 100 if rr[16, 1] else np.nan, (rr[16, 1] - base) / 100 if rr[16, 1] else np.nan]
    rows.append(f(prev) + f(last))
m = np.nanmedian(np.array(rows), axis=0)
for name, o in (("launch i  ", 0), ("launch i+1", 6)):
    print(f"{name}: rollout workgroups enter {m[o]:6.2f} .. {m[o + 1]:6.2f}, exit {m[o + 2]:6.2f} .. {m[o + 3]:6.2f} | aux workgroup (tail of the solve before) enters {m[o + 4]:6.2f}, exits {m[o + 5]:6.2f}")
