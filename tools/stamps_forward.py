"""Timeline of ONE synchronous forward() on the one-launch path (timing build: python tools/build_variant.py timing -DBN_TIMING):
host side -- the forward call, the wait in first_action -- and inside the launch -- the rollout workgroup (0, 0) and the solve's own
tail workgroup on the chip-wide 100 MHz clock.  BN_VARIANT names the build; BN_STATE=cuda hands the state over as a device pointer."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C, numpy as np, torch
from benchnav_amd import build as b
b.LIB_PATH = os.path.join(ROOT, "tools", "_ablate", "lib_%s.so" % os.environ.get("BN_VARIANT", "timing"))
from benchnav_amd import NativeMPPI, synth, _capi
inst = synth.make_instance(256, seed=0)
ref = os.environ.get("BN_REF") == "1"
paced = os.environ.get("BN_PACED") == "1"
_side = torch.cuda.Stream() if os.environ.get("BN_SIDE_STREAM") == "1" else None      # the planner on a stream of torch's other than the default one
pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, stream=(_side.cuda_stream if _side else 0), reference_order=ref, host_paced=paced)
assert pl.host_paced() == paced
pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
stamps = torch.zeros(1024, dtype=torch.int64, device="cuda")
lib = pl._lib
lib.bn_mppi_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
lib.bn_mppi_debug_set_stamps(pl._h, C.c_void_p(stamps.data_ptr()))
st_host = np.ascontiguousarray(inst.start.numpy(), np.float32)
st_dev = inst.start.cuda()
dev_state = os.environ.get("BN_STATE") == "cuda"
fa = np.empty(2, np.float32)
fap = fa.ctypes.data_as(C.POINTER(C.c_float))
fwd = lib.bn_mppi_forward_async if dev_state else lib.bn_mppi_forward_state_async
sptr = C.c_void_p(st_dev.data_ptr() if dev_state else st_host.ctypes.data)
h = pl._h
torch.cuda.synchronize()
for _ in range(200):
    fwd(h, sptr, None, 0, None); lib.bn_mppi_first_action(h, 0, fap)
rows, host, pars = [], [], []
for rep in range(60):
    if paced:
        for _ in range(3):
            fwd(h, sptr, None, 0, None); lib.bn_mppi_first_action(h, 0, fap)      # steady state: the launch of the step timed below is waiting
        time.sleep(20e-6)
    else:
        torch.cuda.synchronize()
    t0 = time.perf_counter(); fwd(h, sptr, None, 0, None); t1 = time.perf_counter(); lib.bn_mppi_first_action(h, 0, fap); t2 = time.perf_counter()
    if paced: lib.bn_mppi_flush(h)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    host.append((t1 - t0, t2 - t1, t3 - t2))
    pars.append(int(pl.solve_count() - 1) & 1)
    rows.append(stamps.cpu().numpy().astype(np.float64).copy())
host = np.median(np.array(host), axis=0) * 1e6
# a loop without synchronisation: the rate the drop-in boundary runs at
n = int(os.environ.get("BN_LOOP", "2000"))
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(n):
    fwd(h, sptr, None, 0, None); lib.bn_mppi_first_action(h, 0, fap)
loop = (time.perf_counter() - t) / n * 1e6
lib.bn_mppi_flush(h)
torch.cuda.synchronize()
R = np.stack(rows)
rel = []
for r, par in zip(R, pars):
    base = 32 + 16 * par
    t0w = r[base + 13] if paced else r[base + 0]          # paced: the moment the rollout workgroup saw "go"; else its start
    rel.append([(r[base + s_] - t0w) / 100.0 for s_ in (0, 14, 1, 2, 3, 4, 5)] + [(r[900 + 16 * par + i] - t0w) / 100.0 for i in range(7)] + [(r[910 + 4 * par] - t0w) / 100.0])
rel = np.median(np.array(rel), axis=0)
print(f"host: forward call {host[0]:.1f} us | first_action wait {host[1]:.1f} us | rest of the kernel (sync) {host[2]:.1f} us | loop forward+first_action {loop:.2f} us/step  (state={'cuda' if dev_state else 'host by value'}, ref_order={ref}, host_paced={paced})")
ref0 = "the rollout wg saw go" if paced else "the rollout wg's start"
print("rollout wg(0,0), us after %s: start %.2f | ready, polls for the request %.2f | prologue end %.2f | chunk0 %.2f | chain end %.2f | after barrier %.2f | column sums out %.2f" % ((ref0,) + tuple(rel[:7])))
print("self tail, us after %s: enter %.2f | window staged %.2f | rows seen %.2f | merged + mailbox %.2f | X* rolled %.2f | all waves %.2f | end %.2f | (paced) request taken from the host %.2f" % ((ref0,) + tuple(rel[7:])))
if paced and hasattr(lib, "bn_mppi_debug_hp_clock"):
    lib.bn_mppi_debug_hp_clock()
pl.close()
