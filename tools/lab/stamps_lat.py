"""Phase timeline of rollout_lat_kernel (timing build: python tools/stamps.py build), pipelined steady state."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes as C, numpy as np, torch
from benchnav_amd import build as b
b.LIB_PATH = os.path.join(ROOT, "tools", "_ablate", "lib_%s.so" % os.environ.get("BN_VARIANT", "timing"))
from benchnav_amd import NativeMPPI, synth
inst = synth.make_instance(256, seed=0)
for kern in os.environ.get("BN_KERNELS", "lat,role").split(","):
    pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, stream=0, kernel=kern)
    pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
    stamps = torch.zeros(64 + 4 * 64, dtype=torch.int64, device="cuda")
    pl._lib.bn_mppi_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
    pl._lib.bn_mppi_debug_set_stamps(pl._h, C.c_void_p(stamps.data_ptr()))
    st = inst.start.cuda(); torch.cuda.synchronize()
    rows = []
    for rep in range(5):
        for _ in range(30): pl.solve_async_device(st.data_ptr())
        torch.cuda.synchronize(); rows.append(stamps.cpu().numpy().astype(np.float64).copy())
    pp = np.median(np.stack(rows), axis=0)
    d = lambda i, j: (pp[j] - pp[i]) / 2400.0
    if kern == "lat":
        print(f"[lat ] prologue {d(0,1):.2f} | chunk0 {d(1,2):.2f} | chain rest {d(2,3):.2f} | -> final barrier {d(3,9):.2f} | cost/exp {d(9,4):.2f} | "
              f"colsum {d(4,5):.2f} | total {d(0,5):.2f} us")
    else:
        print(f"[role] prologue {d(0,1):.2f} | chunk0 {d(1,2):.2f} | chunks {d(2,3):.2f} | cost {d(3,4):.2f} | colsum {d(4,5):.2f} | total {d(0,5):.2f} us")
    pl.close()
