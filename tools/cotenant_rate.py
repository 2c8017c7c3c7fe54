"""Dependent solves per second with ANOTHER PROCESS on the GPU (tools/ubench_bg.bin: workgroups streaming memory): the overlapped
chain alone ran at half the one-stream rate there; since round 6 the handle watches its own cadence and moves to one
stream by itself (bn_mppi_overlap_mode).  Build the hog first: hipcc --offload-arch=gfx950 -O3 tools/ubench_bg.hip -o tools/ubench_bg.bin"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from benchnav_amd import NativeMPPI, synth
inst = synth.make_instance(256, seed=0)
st = inst.start.cuda()
def rate(pl, n=2000, reps=6):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pl.solve_n_async_device(n, st.data_ptr()); pl.flush(); torch.cuda.synchronize()
        out.append(n / (time.perf_counter() - t0)); pl.sync()
    return out
def planner(**kw):
    pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, **kw)
    pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
    return pl
for label, hog in (("alone", None), ("with a co-tenant (4 workgroups streaming memory)", ["4", "6", "0"]), ("with a co-tenant (32 workgroups)", ["32", "6", "0"])):
    bg = subprocess.Popen([os.path.join(ROOT, "tools", "ubench_bg.bin")] + hog, stdout=subprocess.DEVNULL) if hog else None
    if bg: time.sleep(1.5)
    a, b = planner(), planner(overlap=False)
    ra, rb = rate(a), rate(b)
    print(f"{label}: self-protecting handle {[round(x / 1e3, 1) for x in ra]} k solves/s (mode {a.overlap_mode()} at the end) | BN_FLAG_NO_OVERLAP {[round(x / 1e3, 1) for x in rb]}", flush=True)
    a.close(); b.close()
    if bg: bg.wait()
