"""Fixed cost of bench.py's timed region (sync, K dependent solves, last tail, sync) over K: slope = per-solve time, intercept =
first-launch latency + cross-stream fork / join + the last tail kernel + the final synchronisation."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from benchnav_amd import build as _b
if os.environ.get("BN_TOOL_LIB", "main") != "main":                     # a variant built by tools/build_variant_fast.py <name>
    _b.LIB_PATH = os.path.join(ROOT, "tools", "_ablate", "lib_%s.so" % os.environ["BN_TOOL_LIB"])
from benchnav_amd import NativeMPPI, synth
torch.set_num_threads(1)
inst = synth.make_instance(256, seed=0)
st = inst.start.cuda()
stream = torch.cuda.Stream()
for overlap in (True, False):
    pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, stream=stream.cuda_stream, overlap=overlap)
    pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
    pl.solve_n_async_device(300, st.data_ptr()); pl.sync()
    row = []
    for K in (1, 2, 3, 5, 10, 20, 50, 200):
        ts = []
        for _ in range(60):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            pl.solve_n_async_device(K, st.data_ptr()); pl.flush(); torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        row.append((K, statistics.median(ts) * 1e6))
    (k1, t1), (k2, t2) = row[-3], row[-1]
    slope = (t2 - t1) / (k2 - k1)
    print(f"overlap={overlap!s:5s}: " + "  ".join(f"K={k}: {t:6.1f} us" for k, t in row) + f"   | slope {slope:5.2f} us/solve, intercept at K=20: {row[5][1] - 20 * slope:5.1f} us")
    pl.close()
