"""Consumer B's looks at the chain in the overlapped steady state (timing build: tools/build_variant_fast.py --timing <name>):
per look the cycle counter, the chain's progress as seen and B's own position.  BN_VARIANT selects the library."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C, numpy as np, torch
from benchnav_amd import build as b
b.LIB_PATH = os.path.join(ROOT, "tools", "_ablate", "lib_%s.so" % os.environ.get("BN_VARIANT", "timing"))
from benchnav_amd import NativeMPPI, synth
inst = synth.make_instance(256, seed=0)
pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, kernel="lat")
pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
stamps = torch.zeros(320 + 256, dtype=torch.int64, device="cuda")
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
    log = s[320:560].reshape(-1, 2)
    log = log[log[:, 0] > 0]
    w = s[192:192 + 60].reshape(5, 12)
    c0 = w[1, 9] if w[1, 9] else log[0, 0]
    print(f"rep {rep}: {len(log)} looks; chain last step at {(w[1, 11] - c0) / 2400:.2f} us, B out of its loop at {(w[3, 0] - c0) / 2400:.2f}")
    print("  us after the chain's start: seen/own ", " ".join(f"{(c - c0) / 2400:.2f}:{v & 0xffff}/{v >> 16}" for c, v in log))
