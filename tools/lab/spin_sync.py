"""Does the host's wait at the end of bench.py's timed region cost anything?  Region time (sync, K dependent solves, tail, sync) with
the runtime's default wait against hipDeviceScheduleSpin (hipSetDeviceFlags) and against an empty region (sync, sync) and a region
holding one empty-ish kernel (torch's fill of 1 element): what the runtime alone needs for launch + completion."""
import ctypes, os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from benchnav_amd import NativeMPPI, synth
torch.set_num_threads(1)
hip = ctypes.CDLL("libamdhip64.so")
inst = synth.make_instance(256, seed=0)
st = inst.start.cuda()
one = torch.zeros(1, device="cuda")
stream = torch.cuda.Stream()


def med(f, n=200):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e6, ts[len(ts) // 10] * 1e6


for mode in ("default", "spin", "yield", "blocking", "auto"):
    flag = {"default": None, "spin": 1, "yield": 2, "blocking": 4, "auto": 0}[mode]
    if flag is not None:
        rc = hip.hipSetDeviceFlags(ctypes.c_uint(flag))
        print(f"hipSetDeviceFlags({flag}) -> {rc}")
    pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, stream=stream.cuda_stream)
    pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
    pl.solve_n_async_device(300, st.data_ptr()); pl.sync()
    out = [f"{mode:8s}"]
    out.append("empty %.1f/%.1f" % med(lambda: None))
    with torch.cuda.stream(stream):
        out.append("fill %.1f/%.1f" % med(lambda: one.fill_(1.0)))
    for K in (1, 20, 50):
        def region():
            pl.solve_n_async_device(K, st.data_ptr()); pl.flush()
        out.append("K=%d %.1f/%.1f" % ((K,) + med(region, 120)))
    print("  ".join(out) + "   (median / p10, us)")
    pl.close()
