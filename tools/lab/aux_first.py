"""256 instances on the one-wave kernel, aux (tail) workgroups in the LAST grid rows (shipped) against the FIRST (BN_AUX_FIRST, experiment
build: python tools/build_variant.py exp): per-launch time overlapped and on one stream, outputs compared.  VERDICT r5 #3."""
import os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from benchnav_amd import build as b
b.LIB_PATH = os.path.join(ROOT, "tools", "_ablate", "lib_%s.so" % os.environ.get("BN_TOOL_LIB", "exp")) if os.environ.get("BN_TOOL_LIB", "exp") != "main" else b.LIB_PATH
from benchnav_amd import NativeMPPI, synth
inst = synth.make_instance(256, seed=0)
B = int(os.environ.get("BN_BS", "256"))
st = torch.stack([inst.start] * B).cuda()
for overlap in (True, False):
    pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, num_instances=B, shared_map=True, overlap=overlap)
    pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
    pl.solve_n_async_device(60, st.data_ptr()); pl.sync()
    best = None
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pl.solve_n_async_device(300, st.data_ptr()); pl.flush(); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 300; pl.sync()
        best = dt if best is None else min(best, dt)
    h = hashlib.sha256(pl.get_mean(0).tobytes() + pl.weights(B - 1).tobytes()).hexdigest()[:12]
    print(f"lib={os.environ.get('BN_TOOL_LIB', 'exp')} B={B} aux_first={'BN_AUX_FIRST' in os.environ} overlap={overlap}: {best * 1e6:.2f} us per launch, outputs {h}", flush=True)
    pl.close()
