"""Per-workgroup timeline of one batched rollout launch (timing build: python tools/stamps.py build): when each workgroup
started and ended (chip-wide 100 MHz clock), how many cycles it ran, and where (XCC / SE / CU from HW_ID).
    BN_BS=60,64 BN_XCD_PACK=0 python tools/block_trace.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes as C, numpy as np, torch
from benchnav_amd import build as b
b.LIB_PATH = os.path.join(ROOT, "tools", "_ablate", "lib_timing.so")
from benchnav_amd import NativeMPPI, synth
inst = synth.make_instance(256, seed=0)
NBLK = 16
for B in [int(x) for x in os.environ.get("BN_BS", "16,32,60,64,128").split(",")]:
    pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, stream=0, num_instances=B, shared_map=True, kernel="role",
                    lean=bool(int(os.environ.get("BN_LEAN", "0"))))
    pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
    xs = 3 if os.environ.get("BN_XCD_PACK", "0") == "1" else 0
    gx = NBLK << xs
    rows = (B + (1 << xs) - 1) >> xs
    aux_rows = (B + gx - 1) // gx
    nb = gx * (rows + aux_rows)
    stamps = torch.zeros(64 + 4 * nb, dtype=torch.int64, device="cuda")
    pl._lib.bn_mppi_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
    pl._lib.bn_mppi_debug_set_stamps(pl._h, C.c_void_p(stamps.data_ptr()))
    st = torch.stack([inst.start] * B).cuda(); torch.cuda.synchronize()
    for _ in range(20): pl.solve_async_device(st.data_ptr())
    torch.cuda.synchronize()
    r = stamps.cpu().numpy()[64:].reshape(nb, 4).astype(np.int64)
    live = r[:, 1] != 0                                     # idle workgroups return before the trace
    aux = (np.arange(nb) >= gx * rows) & live
    rol = (np.arange(nb) < gx * rows) & live
    base = r[live, 0].min()
    t0 = (r[:, 0] - base) / 100.0; t1 = (r[:, 1] - base) / 100.0; dur = r[:, 2] / 2400.0
    hw = r[:, 3] & 0xffffffff; xcc = (r[:, 3] >> 32) & 0xf
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    place = xcc * 1000 + se * 100 + sh * 10 + cu
    uniq, cnt = np.unique(place[rol], return_counts=True)
    q = lambda v, p: float(np.percentile(v, p))
    print(f"B={B} xs={xs}: {int(rol.sum())} rollout + {int(aux.sum())} aux workgroups, launch span {t1[live].max():.1f} us\n"
          f"   rollout: start med {q(t0[rol],50):.1f} p90 {q(t0[rol],90):.1f} max {t0[rol].max():.1f} | dur med {q(dur[rol],50):.1f} p90 {q(dur[rol],90):.1f} max {dur[rol].max():.1f}"
          f" | end med {q(t1[rol],50):.1f} p90 {q(t1[rol],90):.1f} max {t1[rol].max():.1f}\n"
          f"   aux    : start min {t0[aux].min():.1f} med {q(t0[aux],50):.1f} max {t0[aux].max():.1f} | dur med {q(dur[aux],50):.1f} max {dur[aux].max():.1f} | end med {q(t1[aux],50):.1f} max {t1[aux].max():.1f}\n"
          f"   rollout workgroups per CU: min {cnt.min()} median {int(np.median(cnt))} max {cnt.max()} on {len(uniq)} CUs | per XCC {np.bincount(xcc[rol], minlength=8).tolist()}")
    # CUs per shader engine (harvesting can leave them unequal) against the workgroups each engine was dealt
    se_key = xcc * 10 + se
    per_se = []
    for k in np.unique(se_key[rol]):
        m = rol & (se_key == k)
        per_se.append((len(np.unique(place[m])), int(m.sum()), int((m & (t0 > 5.0)).sum())))
    from collections import Counter
    print("   shader engines by (CUs, rollout workgroups dealt, of those started late):", dict(Counter(per_se)))
    tg = (hw >> 16) & 0xf
    # duration by start rank within the CU (0 = first workgroup the CU received), and whether the slot ids are distinct
    by_rank = {k: [] for k in range(8)}
    distinct = 0
    for pl_ in uniq:
        m = np.where(rol & (place == pl_))[0]
        m = m[np.argsort(r[m, 0], kind="stable")]
        distinct += int(len(set(tg[m].tolist())) == len(m))
        for k, i in enumerate(m):
            by_rank[k].append(dur[i])
    print("   duration by arrival rank on the CU:", " ".join(f"#{k}: {np.median(v):.1f}" for k, v in by_rank.items() if v),
          f"| CUs whose co-resident workgroups have distinct TG_ID: {distinct}/{len(uniq)}")
    hist, edges = np.histogram(t1[rol], bins=np.arange(0, t1[live].max() + 2, 2.0))
    print("   rollout end-time histogram (2 us bins):", " ".join(f"{int(e)}:{h}" for e, h in zip(edges[:-1], hist) if h))
    pl.close()
