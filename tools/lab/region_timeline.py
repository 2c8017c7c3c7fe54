"""Where bench.py's K=20 timed region spends its fixed cost: host stamps (call, enqueue done, sync done) and a HIP event pair
(first packet .. behind the tail) for the closing synchronisation variants.  Env: BN_JOIN=1 restores the cross-stream join."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from benchnav_amd import NativeMPPI, synth
torch.set_num_threads(1)
inst = synth.make_instance(256, seed=0)
st = inst.start.cuda()
stream = torch.cuda.current_stream()
K = int(os.environ.get("K", "20"))
for overlap in (True, False):
    pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, stream=stream.cuda_stream, overlap=overlap)
    pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
    pl.solve_n_async_device(300, st.data_ptr()); pl.sync()
    for closing in ("torch.cuda.synchronize", "stream.synchronize", "pl.sync"):
        rows = []
        for rep in range(80):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            use_ev = rep % 2 == 1
            t0 = time.perf_counter()
            if use_ev: e0.record(stream)
            pl.solve_n_async_device(K, st.data_ptr())
            t1 = time.perf_counter()
            pl.flush()
            if use_ev: e1.record(stream)
            t2 = time.perf_counter()
            if closing == "torch.cuda.synchronize": torch.cuda.synchronize()
            elif closing == "stream.synchronize": stream.synchronize()
            else: pl.sync()
            t3 = time.perf_counter()
            torch.cuda.synchronize()
            rows.append(((t3 - t0) * 1e6, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, e0.elapsed_time(e1) * 1e3 if use_ev else None, use_ev))
        med = lambda i, ev: statistics.median(r[i] for r in rows if r[5] == ev)
        print(f"overlap={overlap!s:5s} close={closing:24s} K={K}: total {med(0, False):6.1f} us (enqueue {med(1, False):5.1f}, flush {med(2, False):4.1f}, wait {med(3, False):6.1f})"
              f"   with events: total {med(0, True):6.1f}, GPU first packet..behind tail {med(4, True):6.1f}")
    pl.close()
