"""Sustained dependent solves (K=1024, T=50, 256x256, Philox) on a given build of the library, by path -- for A/B runs of two builds
in alternating processes on one box:  python tools/lab/ab_headline.py tools/_ablate/lib_r5.so"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from benchnav_amd import _capi, synth
lib = C.CDLL(os.path.abspath(sys.argv[1]))
H = C.c_void_p
lib.bn_mppi_create.argtypes = [C.POINTER(_capi.Config), C.POINTER(H)]
lib.bn_mppi_set_map.argtypes = [H, C.c_int32, C.c_void_p, C.c_int]
lib.bn_mppi_set_goal.argtypes = [H, C.c_int32, C.POINTER(C.c_float)]
lib.bn_mppi_solve_n_async.argtypes = [H, C.c_int32, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int32, C.c_int64]
lib.bn_mppi_sync.argtypes = [H]; lib.bn_mppi_flush.argtypes = [H]; lib.bn_mppi_destroy.argtypes = [H]
lib.bn_last_error.restype = C.c_char_p
cfg = _capi.Config(); lib.bn_mppi_config_init(C.byref(cfg))
G = 256
cfg.horizon, cfg.num_samples, cfg.num_instances, cfg.grid_size, cfg.resolution = 50, 1024, 1, G, 0.5
for i in range(2): cfg.x_limits[i] = cfg.y_limits[i] = (0.0, G * 0.5)[i]
stream = torch.cuda.Stream(); cfg.stream = stream.cuda_stream
h = H(); rc = lib.bn_mppi_create(C.byref(cfg), C.byref(h)); assert rc == 0, lib.bn_last_error()
inst = synth.make_instance(G, seed=0)
risk = inst.risk.numpy(); goal = inst.goal.numpy().astype("float32")
assert lib.bn_mppi_set_map(h, 0, risk.ctypes.data, 0) == 0
assert lib.bn_mppi_set_goal(h, 0, goal.ctypes.data_as(C.POINTER(C.c_float))) == 0
st = inst.start.cuda(); torch.cuda.synchronize()
def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    assert lib.bn_mppi_solve_n_async(h, n, st.data_ptr(), 1, None, 0, 1, 0) == 0
    lib.bn_mppi_flush(h); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    lib.bn_mppi_sync(h); return dt
run(500)
sus = min(run(3000) for _ in range(5)) / 3000 * 1e6
reg = sorted(run(20) for _ in range(300))
print(f"{os.path.basename(sys.argv[1]):28s} sustained {sus:.3f} us/solve ({1e6 / sus:.0f}/s) | 20-solve region median {reg[150] * 1e6:.1f} us, p10 {reg[30] * 1e6:.1f} us -> {20 / reg[150]:.0f}/s")
lib.bn_mppi_destroy(h)
