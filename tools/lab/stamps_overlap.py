"""Timeline of the overlapped steady state (timing build: python tools/stamps.py build): wall-clock stamps (100 MHz, chip-wide)
of workgroup 0 of two consecutive launches, kept by solve parity."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes as C, numpy as np, torch
from benchnav_amd import build as b
b.LIB_PATH = os.path.join(ROOT, "tools", "_ablate", "lib_%s.so" % os.environ.get("BN_VARIANT", "timing"))
from benchnav_amd import NativeMPPI, synth
inst = synth.make_instance(256, seed=0)
lean = bool(int(os.environ.get("BN_LEAN", "0")))
T_ = int(os.environ.get("BN_T", "50"))
pl = NativeMPPI(horizon=T_, num_samples=1024, grid_size=256, resolution=0.5, kernel="lat", lean=lean)
pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
stamps = torch.zeros(64 + 4 * 64, dtype=torch.int64, device="cuda")
pl._lib.bn_mppi_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
pl._lib.bn_mppi_debug_set_stamps(pl._h, C.c_void_p(stamps.data_ptr()))
st = inst.start.cuda(); torch.cuda.synchronize()
rows, wrows = [], []
for rep in range(15):
    pl.solve_n_async_device(201, st.data_ptr()); pl.sync()
    s = stamps.cpu().numpy().astype(np.float64)
    wrows.append(s[192:192 + 60].reshape(5, 12).copy())
    n = pl.solve_count()                      # the last launch computed solve n-1
    last, prev = s[32 + ((n - 1) & 1) * 16:][:16], s[32 + ((n - 2) & 1) * 16:][:16]
    rows.append(np.concatenate([(last - last[0]) / 100.0, [(last[13] - prev[5]) / 100.0, (last[5] - prev[5]) / 100.0, (last[0] - prev[0]) / 100.0]]))
r = np.median(np.stack(rows), axis=0)
print(f"entry->granules seen {r[13]:.2f} | merge {r[14] - r[13]:.2f} | chunk0/1 controls {r[1] - r[14]:.2f} | chunk 0 {r[2] - r[1]:.2f} | chain rest {r[3] - r[2]:.2f} | "
      f"final barrier {r[9] - r[3]:.2f} | cost/exp {r[4] - r[9]:.2f} | colsum+granules {r[5] - r[4]:.2f}")
print(f"granules stored (prev) -> seen (next) {r[16]:.2f} us | period {r[17]:.2f} us | entry-to-entry {r[18]:.2f} us | seen->stored {r[5] - r[13]:.2f} us")
w = np.median(np.stack(wrows), axis=0)
names = {7: "granules seen", 8: "merged (before barrier)", 9: "after barrier", 0: "at final barrier", 1: "after final barrier", 2: "cost done (wave 4)",
         3: "at e barrier", 4: "after e barrier", 5: "column sums issued", 6: "published", 10: "chain: per-step phase / producers: done producing", 11: "chain: last step / producers: second job done"}
t0 = w[:, 7][w[:, 7] > 0].min()          # (the fast prologue's consumers never look at the granules)
print("per-wave cycle stamps of workgroup 0, us after the first wave saw its granules (waves: 0 consumer A, 1 chain, 2-3 producers, 4 consumer B)")
for i in (7, 8, 9, 10, 11, 0, 1, 2, 3, 4, 5, 6):
    print(f"  {names[i]:26s}", "  ".join(f"{(w[k, i] - t0) / 2400.0:7.2f}" if w[k, i] else "      -" for k in range(5)))
