import sys, os, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import differential as D
from benchnav_amd import _capi
seed = 2681
c = D.case(seed + 300_000); c["noise"] = "philox"
B, K, T = c["B"], c["K"], c["T"]
st = torch.from_numpy(c["states"]).cuda(); torch.cuda.synchronize()
def run(knobs, script):
    with D.make(c, **knobs) as pl:
        for kind, m in script:
            if kind == "batch": pl.solve_n_async_device(m, st.data_ptr())
            elif kind == "single": pl.solve_async_device(st.data_ptr())
            elif kind == "expire": _capi.check(pl._lib.bn_mppi_debug_expire_wait(pl._h))
            elif kind == "sync": pl.sync()
        return D.outputs(pl, c, knobs.get("lean", False)), pl.recovery_count()
for nplain, script in ((23, [("batch", 16), ("single", 1), ("single", 1), ("batch", 5), ("expire", 0)]),
                       (21, [("batch", 16), ("batch", 5), ("expire", 0)]),
                       (17, [("batch", 16), ("single", 1), ("expire", 0)]),
                       (16, [("batch", 16), ("expire", 0)]),
                       (22, [("batch", 16), ("single", 1), ("batch", 5), ("expire", 0)])):
    want, _ = run(dict(overlap=False), [("single", 1)] * nplain)
    for lean in (True, False):
        got, rec = run(dict(c["knobs"], lean=lean), script)
        d = [k for k, v in got.items() if not np.array_equal(v, want[k], equal_nan=True)]
        print(script, "lean", lean, "recoveries", rec, "diff", d, flush=True)
