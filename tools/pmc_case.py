"""One planner configuration, ~200 dependent launches, nothing else: what the PMC passes of tools/profile_r2.sh profile so
that every counter row belongs to exactly one bench leg.   python tools/pmc_case.py B64_lean"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from benchnav_amd import NativeMPPI, synth

case = sys.argv[1]
n = int(os.environ.get("BN_N", 200))
G, K, T, RES = 256, 1024, 50, 0.5
lean = case.endswith("_lean")
base = case.replace("_lean", "")
torch.set_num_threads(1)
if base in ("B1", "B8", "B64", "B256"):
    B = int(base[1:])
    insts = [synth.make_instance(G, seed=s, resolution=RES, jitter=B > 1) for s in range(min(B, 64))]
    shared = B > 64 or B == 1
    pl = NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=RES, num_instances=B, shared_map=shared, stream=0, lean=lean)
    if shared:
        pl.set_map(insts[0].risk.numpy()); pl.set_goal(insts[0].goal.numpy())
        st = torch.stack([insts[0].start] * B).cuda()
    else:
        for b, it in enumerate(insts):
            pl.set_map(it.risk.numpy(), b); pl.set_goal(it.goal.numpy(), b)
        st = torch.stack([it.start for it in insts]).cuda()
elif base == "sampled":
    inst = synth.make_instance(G, seed=0, resolution=RES)
    pl = NativeMPPI(horizon=T, num_samples=8192, grid_size=G, resolution=RES, sampled_slip=True, stream=0)
    pl.set_map(inst.risk.numpy()); pl.set_slip_std(synth.slip_std_map(G, seed=0).numpy()); pl.set_goal(inst.goal.numpy())
    st = inst.start.cuda()
elif base == "ref5000":
    from benchnav_amd.risk import infer_risk_map
    risk = infer_risk_map(synth.smooth_risk_map(64, 9) * 0.7, synth.slip_std_map(64, 9), "cvar", 0.9, seed=0).cpu().numpy()
    pl = NativeMPPI(horizon=T, num_samples=5000, grid_size=64, resolution=RES, stream=0, lean=lean)
    pl.set_map(risk); pl.set_goal([24.0, 24.0])
    st = torch.tensor([8.0, 8.0, 0.7853981633974483], device="cuda")
elif base == "c5":
    inst = synth.make_instance(512, seed=0, resolution=RES)
    pl = NativeMPPI(horizon=100, num_samples=16384, grid_size=512, resolution=RES, stream=0, lean=lean)
    pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
    st = inst.start.cuda()
else:
    raise SystemExit(f"unknown case {case}")
torch.cuda.synchronize()
pl.solve_n_async_device(n, st.data_ptr())
pl.sync()
pl.close()
print("done", case, n)
