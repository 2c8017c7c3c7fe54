"""Per-workgroup timeline of TWO consecutive launches of an overlapped batch of the role kernel (timing build:
python tools/stamps.py build): start / end of every workgroup on the chip-wide clock, rows kept by solve parity.
    BN_BS=40,64 python tools/block_trace_overlap.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes as C, numpy as np, torch
from benchnav_amd import build as b
b.LIB_PATH = os.path.join(ROOT, "tools", "_ablate", "lib_timing.so")
from benchnav_amd import NativeMPPI, synth
inst = synth.make_instance(256, seed=0)
NBLK = 16
for B in [int(x) for x in os.environ.get("BN_BS", "40,64").split(",")]:
    pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, num_instances=B, shared_map=True, kernel=os.environ.get("BN_KERNEL", "role"))
    pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
    nb = NBLK * (B + (B + NBLK - 1) // NBLK)
    stamps = torch.zeros(64 + 4 * 2 * nb, dtype=torch.int64, device="cuda")
    pl._lib.bn_mppi_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
    pl._lib.bn_mppi_debug_trace_by_parity.argtypes = [C.c_void_p, C.c_int]
    pl._lib.bn_mppi_debug_set_stamps(pl._h, C.c_void_p(stamps.data_ptr()))
    pl._lib.bn_mppi_debug_trace_by_parity(pl._h, 1)
    st = torch.stack([inst.start] * B).cuda(); torch.cuda.synchronize()
    n = 41
    if os.environ.get("BN_EPISODE"):
        lat = inst.risk.numpy()[None]; pl.env_attach(lat, np.full_like(lat, 0.05))
        pl.episode(n, np.stack([inst.start.numpy()] * B))
        pl.episode(n, np.stack([inst.start.numpy()] * B))
    else:
        pl.solve_n_async_device(n, st.data_ptr()); pl.sync()
    total = pl.solve_count()
    r = stamps.cpu().numpy()[64:].reshape(2, nb, 4).astype(np.int64)
    last, prev = r[(total - 1) & 1], r[(total - 2) & 1]          # the final launch and the one before it
    idx = np.arange(nb)
    rol = idx < NBLK * B
    aux = (idx >= NBLK * B) & (last[:, 1] != 0)
    base = prev[rol, 0].min()
    q = lambda v, p: float(np.percentile(v, p))
    for name, rr in (("launch i  ", prev), ("launch i+1", last)):
        t0 = (rr[:, 0] - base) / 100.0; t1 = (rr[:, 1] - base) / 100.0
        print(f"B={B} {name}: rollout start min {t0[rol].min():6.1f} med {q(t0[rol],50):6.1f} max {t0[rol].max():6.1f} | end min {t1[rol].min():6.1f} med {q(t1[rol],50):6.1f} max {t1[rol].max():6.1f}"
              f" | aux start {t0[aux].min():6.1f}..{t0[aux].max():6.1f} end {t1[aux].min():6.1f}..{t1[aux].max():6.1f}")
        # per instance: when its first / last rollout workgroup started, when its last one ended
        s_b = t0[rol].reshape(B, NBLK); e_b = t1[rol].reshape(B, NBLK)
        pick = [0, B // 4, B // 2, 3 * B // 4, B - 1]
        print("      instance: " + "  ".join(f"{b_:3d}: start {s_b[b_].min():5.1f}-{s_b[b_].max():5.1f} end {e_b[b_].max():5.1f}" for b_ in pick))
    pl.close()
