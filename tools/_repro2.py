import sys, os, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import numpy as np, torch
from benchnav_amd import build as _b
if os.environ.get("BN_TOOL_LIB"): _b.LIB_PATH = os.path.join("/root/repo/tools/_ablate", "lib_%s.so" % os.environ["BN_TOOL_LIB"])
import fuzz_features as F
from benchnav_amd import _capi
seed = 2362
c = F.case(seed)
st = torch.from_numpy(c["states"]).cuda(); torch.cuda.synchronize()
variants = {
    "orig": c["cuts"],
    "1fa+11": [(1, "first_action"), (11, "none")],
    "2fa+11": [(2, "first_action"), (11, "none")],
}
reps = int(sys.argv[1])
for lean in (True,):
    for name, cuts in variants.items():
        rec = 0; t0 = time.time()
        for i in range(reps):
            kn = dict(c["knobs"], lean=lean)
            with F.make(c, **kn) as pl:
                for m, then in cuts:
                    pl.solve_n_async_device(m, st.data_ptr()) if m > 1 else pl.solve_async_device(st.data_ptr())
                    if then == "weights": pl.weights(0)
                    elif then == "first_action": pl.first_action(c["B"] - 1)
                    elif then == "sync": pl.sync()
                pl.sync()
                rec += pl.recovery_count()
        print(f"lean={lean} {name:8s}: {rec} recoveries in {reps} handles ({time.time() - t0:.1f}s)", flush=True)
