"""Per-workgroup timeline of one batched rollout launch (timing build): when each workgroup started and ended
(chip-wide 100 MHz clock), how many cycles it ran, and where (XCC / SE / CU from HW_ID)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C, numpy as np, torch
from benchnav_amd import build as b
b.LIB_PATH = os.path.join(ROOT, "tools", "_ablate", "lib_timing.so")
from benchnav_amd import NativeMPPI, synth
inst = synth.make_instance(256, seed=0)
for B in (16, 32, 64, 128):
    pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, stream=0, num_instances=B, shared_map=True)
    pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
    nb = 17 * B
    stamps = torch.zeros(64 + 4 * nb, dtype=torch.int64, device="cuda")
    pl._lib.bn_mppi_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
    pl._lib.bn_mppi_debug_set_stamps(pl._h, C.c_void_p(stamps.data_ptr()))
    st = torch.stack([inst.start] * B).cuda(); torch.cuda.synchronize()
    for _ in range(20): pl.solve_async_device(st.data_ptr())
    torch.cuda.synchronize()
    r = stamps.cpu().numpy()[64:].reshape(nb, 4).astype(np.int64)
    t0 = (r[:, 0] - r[:, 0].min()) / 100.0; t1 = (r[:, 1] - r[:, 0].min()) / 100.0; dur = r[:, 2] / 2400.0
    aux = (np.arange(nb) % 17) == 16
    hw = r[:, 3] & 0xffffffff; xcc = (r[:, 3] >> 32) & 0xf
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    place = xcc * 1000 + se * 100 + sh * 10 + cu
    uniq, cnt = np.unique(place, return_counts=True)
    print(f"B={B}: {nb} workgroups  launch span {t1.max():.1f} us | start times: median {np.median(t0):.1f} p90 {np.percentile(t0,90):.1f} max {t0.max():.1f} | "
          f"rollout wg duration median {np.median(dur[~aux]):.1f} p90 {np.percentile(dur[~aux],90):.1f} max {dur[~aux].max():.1f} | aux median {np.median(dur[aux]):.1f} max {dur[aux].max():.1f} | "
          f"distinct CUs {len(uniq)}  workgroups per CU: min {cnt.min()} median {int(np.median(cnt))} max {cnt.max()} | per XCC {np.bincount(xcc, minlength=8).tolist()}")
    pl.close()
