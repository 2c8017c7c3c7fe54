import sys, os, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from benchnav_amd import build as _b
if os.environ.get("BN_TOOL_LIB"): _b.LIB_PATH = os.path.join("/root/repo/tools/_ablate", "lib_%s.so" % os.environ["BN_TOOL_LIB"])
import numpy as np, torch
import differential as D
seed = 2681
c = D.case(seed + 300_000); c["noise"] = "philox"
if os.environ.get("NOREF"): c["common"]["u_max"] = [1.0, 1.0]; c["common"]["reference_order"] = False
st = torch.from_numpy(c["states"]).cuda(); torch.cuda.synchronize()
def run(knobs, script):
    with D.make(c, **knobs) as pl:
        for kind, m in script:
            if kind == "batch": pl.solve_n_async_device(m, st.data_ptr())
            else: pl.solve_async_device(st.data_ptr())
        return D.outputs(pl, c, knobs.get("lean", False)), pl.recovery_count()
script = [("batch", 16), ("single", 1), ("single", 1), ("batch", 5)]
want, _ = run(dict(overlap=False), [("single", 1)] * 23)
reps = int(sys.argv[1]); bad = 0; recs = 0
for i in range(reps):
    if i % 3 == 0: run(dict(overlap=False), [("single", 1)] * 3)          # something else in between
    got, rec = run(c["knobs"], script)
    d = [k for k, v in got.items() if not np.array_equal(v, want[k], equal_nan=True)]
    recs += rec
    if d: bad += 1; print(i, "diff", d, "rec", rec, flush=True)
print("bad", bad, "recoveries", recs, "of", reps)
