"""Expired waits for real: a library built with -DBN_EXPERIMENTS and BN_ALIGN_BIG=1 starts even 70-instance batches on the cold internal
stream again, so that solve 1 overtakes solve 0 now and then, the bounded waits expire and the batches are re-run on
one stream.  Every handle's results after the repair are compared with a handle that never overlapped: they must be equal -- in
particular no NaN may survive (round 4: a starved first launch used to snapshot a mean that later, spoiled solves had already
overwritten).    python tools/build_variant_fast.py stress && BN_ALIGN_BIG=1 BN_TOOL_LIB=stress python tools/recovery_stress.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from benchnav_amd import build as _b
if os.environ.get("BN_TOOL_LIB", "main") != "main":
    _b.LIB_PATH = os.path.join(ROOT, "tools", "_ablate", "lib_%s.so" % os.environ["BN_TOOL_LIB"])
from benchnav_amd import NativeMPPI, synth
torch.set_num_threads(1)
G, K, T, B = 256, 1024, 50, 70
inst = synth.make_instance(G, seed=21)
st = torch.stack([inst.start + torch.tensor([0.01 * b, 0.0, 0.0]) for b in range(B)]).cuda()
torch.cuda.synchronize()
def run(overlap, lean):
    with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, num_instances=B, shared_map=True, seed=5, lean=lean, overlap=overlap, kernel="role",
                    stream=torch.cuda.current_stream().cuda_stream) as pl:       # the null stream: hot from the copies in front (see tools/handle_sequence.py)
        pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
        pl.solve_n_async_device(6, st.data_ptr())
        pl.solve_n_async_device(8, st.data_ptr())
        pl.sync()
        return [(pl.costs(b), pl.weights(b), pl.get_mean(b)) for b in (0, B // 2, B - 1)], pl.recovery_count()
bad = rec = 0
for lean in (True, False):
    ref, _ = run(False, lean)
    for rep in range(int(os.environ.get("REPS", "12"))):
        got, r = run(True, lean)
        rec += r
        ok = all(np.array_equal(a_, c_) for a, c in zip(got, ref) for a_, c_ in zip(a, c))
        bad += not ok
        if not ok:
            print(f"lean={lean} rep {rep}: MISMATCH after {r} recovery(ies); NaN in costs: {bool(np.isnan(got[0][0]).any())}", flush=True)
print(f"handles with expired waits: {rec}, handles with wrong results: {bad}")
