"""Consumer B against the chain in the overlapped steady state (timing build: tools/build_variant_fast.py --timing <name>): when the chain
started chunks 3 / 7 / the last ones, and when consumer B had the slots below 16 / 32 / 44 / 48 behind it.  BN_VARIANT selects the library."""
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
stamps = torch.zeros(320 + 256 + 16, dtype=torch.int64, device="cuda")
pl._lib.bn_mppi_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
pl._lib.bn_mppi_debug_set_stamps(pl._h, C.c_void_p(stamps.data_ptr()))
st = inst.start.cuda(); torch.cuda.synchronize()
for rep in range(3):
    stamps.zero_(); pl.solve_n_async_device(201, st.data_ptr()); pl.sync()
    s = stamps.cpu().numpy()
    w = s[192:192 + 60].reshape(5, 12)
    tail = s[560:576]
    print("  chain tail chunks (us after chain start):", " ".join(f"{i}:{(c - (w[1, 9] or c)) / 2400:.2f}" for i, c in enumerate(tail) if c),
          "| per-step phase", f"{(w[1, 10] - w[1, 9]) / 2400:.2f}", "last step", f"{(w[1, 11] - w[1, 9]) / 2400:.2f}")
    bm = s[544:552].reshape(4, 2)
    print("  chain at the start of chunk:", " ".join(f"{i}:{(c - w[1, 9]) / 2400:.2f}" for i, c in enumerate(tail) if c),
          "| B had slots below .. behind it:", " ".join(f"{int(v)}:{(c - w[1, 9]) / 2400:.2f}" for c, v in bm if c), f"| chain last step {(w[1, 11] - w[1, 9]) / 2400:.2f}, B out {(w[3, 0] - w[1, 9]) / 2400:.2f}")
    print("  SIMD of waves 0..4:", " ".join(str((int(h) >> 4) & 3) for h in s[576:581]), "| CU", " ".join(str((int(h) >> 8) & 15) for h in s[576:581]))
