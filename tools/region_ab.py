"""One line per process: bench.py's timed region (sync, K dependent solves, tail, sync) at K = 20 (and 21: odd batches end
differently), median / p10 / p90 over 300 regions.  For A/B runs of experiment switches over fresh processes (the spread BETWEEN
processes is larger than the spread inside one)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from benchnav_amd import build as _b
if os.environ.get("BN_TOOL_LIB", "main") != "main":
    _b.LIB_PATH = os.path.join(ROOT, "tools", "_ablate", "lib_%s.so" % os.environ["BN_TOOL_LIB"])
from benchnav_amd import NativeMPPI, synth
torch.set_num_threads(1)
inst = synth.make_instance(256, seed=0)
st = inst.start.cuda()
stream = torch.cuda.Stream()
pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, stream=stream.cuda_stream)
pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
pl.solve_n_async_device(300, st.data_ptr()); pl.sync()
out = []
for K in (20, 21):
    ts = []
    for _ in range(300):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pl.solve_n_async_device(K, st.data_ptr()); pl.flush(); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    out.append("K=%d %.1f / %.1f / %.1f" % (K, ts[150] * 1e6, ts[30] * 1e6, ts[270] * 1e6))
print("  ".join(out) + "  us (median / p10 / p90)")
pl.close()
