"""In-kernel phase timing (s_memtime stamps) of the rollout and finish kernels; needs a -DBN_TIMING build:
   python tools/stamps.py build   (here)      python tools/stamps.py   (GPU box)"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "_ablate", "lib_%s.so" % os.environ.get("BN_VARIANT", "timing"))
if len(sys.argv) > 1 and sys.argv[1] == "build":
    from benchnav_amd import build as b
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call([b.hipcc(), *b.HIPCC_FLAGS, "-DBN_TIMING", "-x", "hip", *[os.path.join(b.CSRC, s) for s in b.SOURCES], "-o", LIB])
    print("built", LIB); sys.exit(0)
import ctypes as C, numpy as np, torch
from benchnav_amd import build as b
b.LIB_PATH = LIB
from benchnav_amd import NativeMPPI, _capi, synth
inst = synth.make_instance(256, seed=0)
for noise in ("philox", "t2k"):
    pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, stream=0)
    pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
    stamps = torch.zeros(32, dtype=torch.int64, device="cuda")
    pl._lib.bn_mppi_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
    pl._lib.bn_mppi_debug_set_stamps(pl._h, C.c_void_p(stamps.data_ptr()))
    st = inst.start.cuda(); eps = torch.randn(50, 2, 1024, device="cuda"); torch.cuda.synchronize()
    acc = []
    for _ in range(20):
        if noise == "philox": pl.solve_async_device(st.data_ptr())
        else: pl.solve_async_device(st.data_ptr(), eps.data_ptr(), _capi.BN_NOISE_DEVICE_T2K)
        pl.sync(); acc.append(stamps.cpu().numpy().copy())
    # pipelined steady state: back-to-back async solves, stamps of the last launch (prologue includes the merge)
    for _ in range(30):
        if noise == "philox": pl.solve_async_device(st.data_ptr())
        else: pl.solve_async_device(st.data_ptr(), eps.data_ptr(), _capi.BN_NOISE_DEVICE_T2K)
    torch.cuda.synchronize(); pp = stamps.cpu().numpy().astype(np.float64)
    print(f"[{noise}] pipelined launch: prologue(stage+merge+first controls) {(pp[1]-pp[0])/2400:.2f} | chunk0 {(pp[2]-pp[1])/2400:.2f} | chunks {(pp[3]-pp[2])/2400:.2f} | cost {(pp[4]-pp[3])/2400:.2f} | colsum {(pp[5]-pp[4])/2400:.2f} | total {(pp[5]-pp[0])/2400:.2f} us")
    print(f"[{noise}]   busy us per role in the steady-state loop (chain, producer, producer, consumer A, consumer B):", " ".join(f"{pp[16+i]/2400:.2f}" for i in range(5)))
    pl.sync()
    a = np.stack(acc[5:]).astype(np.float64)
    d = lambda i, j: np.median(a[:, j] - a[:, i]) / 2400.0      # us at 2.4 GHz
    print(f"[{noise}] rollout: prologue(stage window+mean) {d(0,1):.2f} | step0 {d(1,2):.2f} | steps 1..T-1 {d(2,3):.2f} | cost+exp {d(3,4):.2f} | column sums {d(4,5):.2f} | total {d(0,5):.2f} us")
    print(f"[{noise}] finish : stage {d(8,9):.2f} | merge {d(9,10):.2f} | X* rollout {d(10,11):.2f} | total {d(8,11):.2f} us")
    pl.close()
