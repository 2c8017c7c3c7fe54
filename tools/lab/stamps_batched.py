"""In-kernel phase timing under load: B instances per launch (needs tools/_ablate/lib_timing.so)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes as C, numpy as np, torch
from benchnav_amd import build as b
b.LIB_PATH = os.path.join(ROOT, "tools", "_ablate", "lib_timing.so")
from benchnav_amd import NativeMPPI, _capi, synth
inst = synth.make_instance(256, seed=0)
for B in (1, 4, 16, 32, 64, 128):
  for noise in ("philox", "t2k"):
    pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, stream=0, num_instances=B, shared_map=True, profile=True)
    pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
    stamps = torch.zeros(16, dtype=torch.int64, device="cuda")
    pl._lib.bn_mppi_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
    pl._lib.bn_mppi_debug_set_stamps(pl._h, C.c_void_p(stamps.data_ptr()))
    st = torch.stack([inst.start] * B).cuda(); eps = torch.randn(B, 50, 2, 1024, device="cuda"); torch.cuda.synchronize()
    for _ in range(60):
        if noise == "philox": pl.solve_async_device(st.data_ptr())
        else: pl.solve_async_device(st.data_ptr(), eps.data_ptr(), _capi.BN_NOISE_DEVICE_T2K)
    torch.cuda.synchronize(); pp = stamps.cpu().numpy().astype(np.float64)
    km = pl.kernel_ms()
    print(f"B={B:3d} {noise:6s}: block0 prologue {(pp[1]-pp[0])/2400:.2f} | chunk0 {(pp[2]-pp[1])/2400:.2f} | chunks {(pp[3]-pp[2])/2400:.2f} | cost {(pp[4]-pp[3])/2400:.2f} | colsum {(pp[5]-pp[4])/2400:.2f} | total {(pp[5]-pp[0])/2400:.2f} us ; launch mean {km[0]*1e3:.1f} us -> {B/km[0]/1e3:.0f}k solves/s", flush=True)
    pl.close()
