"""In-kernel phase timing of the sampled-slip kernels (config 3); needs the -DBN_TIMING build of tools/stamps.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes as C, numpy as np, torch
from benchnav_amd import build as b
b.LIB_PATH = os.path.join(ROOT, "tools", "_ablate", "lib_%s.so" % os.environ.get("BN_VARIANT", "timing"))
from benchnav_amd import NativeMPPI, synth
inst = synth.make_instance(256, seed=0)
for K in (8192, 1024):
    pl = NativeMPPI(horizon=50, num_samples=K, grid_size=256, resolution=0.5, stream=0, sampled_slip=True)
    pl.set_map(inst.risk.numpy()); pl.set_slip_std(synth.slip_std_map(256, 0).numpy()); pl.set_goal(inst.goal.numpy())
    stamps = torch.zeros(32, dtype=torch.int64, device="cuda")
    pl._lib.bn_mppi_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
    pl._lib.bn_mppi_debug_set_stamps(pl._h, C.c_void_p(stamps.data_ptr()))
    st = inst.start.cuda(); torch.cuda.synchronize()
    acc = []
    for _ in range(20):
        pl.solve_async_device(st.data_ptr()); pl.sync(); acc.append(stamps.cpu().numpy().copy())
    a = np.stack(acc[5:]).astype(np.float64)
    d = lambda i, j: np.median(a[:, j] - a[:, i]) / 2400.0     # us at 2.4 GHz (s_memtime)
    print(f"K={K} rollout: stage+mean {d(0,1):.2f} (incl. draws) | chain {d(1,2):.2f} | stage costs+stores {d(2,3):.2f} | cost+colsum {d(3,5):.2f} | total {d(0,5):.2f} us  block0 end -> last block ticket {d(5,6):.2f} | its merge {d(6,7):.2f}")
    print(f"K={K} finish : stage {d(8,9):.2f} | merge {d(9,10):.2f} | draws+window {d(10,12):.2f} | X* chain {d(12,11):.2f} | total {d(8,11):.2f} us   gap rollout-end -> finish-start {d(5,8):.2f}")
    pl.close()
