"""A few K=20 timed regions (bench.py's contract) for `rocprofv3 --kernel-trace`: prints the host time of every region; the
trace gives the begin / end of every kernel, so  host total - (first begin .. last end)  = submission latency + wake-up."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from benchnav_amd import NativeMPPI, synth
torch.set_num_threads(1)
inst = synth.make_instance(256, seed=0)
st = inst.start.cuda()
stream = torch.cuda.current_stream()
K = int(os.environ.get("K", "20"))
pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, stream=stream.cuda_stream, overlap=not os.environ.get("NO_OVERLAP"))
pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
pl.solve_n_async_device(300, st.data_ptr()); pl.sync()
for rep in range(12):
    torch.cuda.synchronize()
    time.sleep(0.002)                 # a visible gap between regions in the trace
    t0 = time.perf_counter()
    pl.solve_n_async_device(K, st.data_ptr()); pl.flush(); torch.cuda.synchronize()
    print(f"region {rep}: host {1e6 * (time.perf_counter() - t0):.1f} us")
pl.close()
