"""Consumer B's looks over the end of the chain (timing build: tools/build_variant_fast.py --timing <name>): every look from slot 36 on as
(time after the chain's start, slot reached, slots found), next to the chain's chunk starts and its last step.  BN_VARIANT selects the library."""
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
for rep in range(3):
    stamps.zero_(); pl.solve_n_async_device(201, st.data_ptr()); pl.sync()
    s = stamps.cpu().numpy()
    w = s[192:192 + 60].reshape(5, 12)
    c0 = w[1, 9]
    print("chain: chunk starts", " ".join(f"{i}:{(c - c0) / 2400:.2f}" for i, c in enumerate(s[560:576]) if c), f"| last step {(w[1, 11] - c0) / 2400:.2f} | B out {(w[3, 0] - c0) / 2400:.2f}, cost {(w[3, 2] - c0) / 2400:.2f}, at e barrier {(w[3, 3] - c0) / 2400:.2f}; others at e barrier",
          " ".join(f"{(w[i, 3] - c0) / 2400:.2f}" for i in (0, 1, 2, 4)), f"| after {(w[3, 4] - c0) / 2400:.2f} | column sums {(w[3, 5] - c0) / 2400:.2f} | published {(w[3, 6] - c0) / 2400:.2f}")
    tr = s[640:640 + 120].reshape(40, 3)
    print("   B looks (us: slot reached / found):", " ".join(f"{(c - c0) / 2400:.2f}:{int(t)}/{int(n)}" for c, t, n in tr if c))
