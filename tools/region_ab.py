"""A/B of an experiment switch (environment variable read per batch by an experiment build: python tools/build_variant_fast.py exp;
BN_TOOL_LIB=exp) on bench.py's timed region (sync, K dependent solves, tail, sync), ALTERNATING inside one process -- the spread
between processes (+-4 us at K = 20) is larger than most effects worth looking for.
    BN_TOOL_LIB=exp python tools/region_ab.py BN_SELF_TAIL"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from benchnav_amd import build as _b
if os.environ.get("BN_TOOL_LIB", "main") != "main":
    _b.LIB_PATH = os.path.join(ROOT, "tools", "_ablate", "lib_%s.so" % os.environ["BN_TOOL_LIB"])
from benchnav_amd import NativeMPPI, synth
torch.set_num_threads(1)
switch = sys.argv[1] if len(sys.argv) > 1 else "BN_SELF_TAIL"
inst = synth.make_instance(256, seed=0)
st = inst.start.cuda()
stream = torch.cuda.Stream()
pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, stream=stream.cuda_stream)
pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
pl.solve_n_async_device(300, st.data_ptr()); pl.sync()
for K in (20, 21, 50):
    ts = {0: [], 1: []}
    for i in range(600):
        on = i & 1
        if on: os.environ[switch] = "1"
        else: os.environ.pop(switch, None)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pl.solve_n_async_device(K, st.data_ptr()); pl.flush(); torch.cuda.synchronize()
        ts[on].append(time.perf_counter() - t0)
    med = lambda v: sorted(v)[len(v) // 2] * 1e6
    p10 = lambda v: sorted(v)[len(v) // 10] * 1e6
    print(f"K={K}: off {med(ts[0]):.1f} (p10 {p10(ts[0]):.1f})   {switch} on {med(ts[1]):.1f} (p10 {p10(ts[1]):.1f})   difference {med(ts[1]) - med(ts[0]):+.2f} us", flush=True)
os.environ.pop(switch, None)
pl.close()
