"""Shader clock INSIDE the latency kernel (timing build: python tools/build_variant_fast.py --timing timing): the chain wave stamps the
cycle counter and the 100 MHz wall clock behind its first chunk and behind its last step; their ratio is the clock the chain ran at.
Round 4: 2.38-2.43 GHz -- the kernel's waves are not slowed by a power state.  BN_VARIANT=<name> picks the library."""
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
stamps = torch.zeros(64 + 4 * 64 + 512, dtype=torch.int64, device="cuda")
pl._lib.bn_mppi_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
pl._lib.bn_mppi_debug_set_stamps(pl._h, C.c_void_p(stamps.data_ptr()))
st = inst.start.cuda(); torch.cuda.synchronize()
r=[]
for rep in range(15):
    pl.solve_n_async_device(201, st.data_ptr()); pl.sync()
    s = stamps.cpu().numpy().astype(np.float64)
    n = pl.solve_count(); par = (n - 1) & 1
    w2, w3 = s[32 + par*16 + 2], s[32 + par*16 + 3]
    c2, c3 = s[2], s[3]
    if w3 > w2: r.append(((c3-c2)/((w3-w2)*0.01), (w3-w2)*0.01))
print(os.environ.get("BN_VARIANT"), "shader clock during the chain (MHz), chain rest (us):", [ (round(a), round(b_,2)) for a,b_ in r[:8]])
