import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from benchnav_amd import build as _b
if os.environ.get("BN_TOOL_LIB"): _b.LIB_PATH = os.path.join("/root/repo/tools/_ablate", "lib_%s.so" % os.environ["BN_TOOL_LIB"])
import numpy as np, torch
import differential as D
c = D.case(2681 + 300_000); c["noise"] = "philox"
c["common"]["u_max"] = [1.0, 1.0]; c["common"]["reference_order"] = False
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    if k in ("B",): c["B"] = int(v); c["common"]["num_instances"] = int(v); c["maps"] = np.repeat(c["maps"][:1], int(v), 0); c["states"] = np.repeat(c["states"][:1], int(v), 0); c["goals"] = np.repeat(c["goals"][:1], int(v), 0)
    elif k == "K": c["K"] = int(v); c["common"]["num_samples"] = int(v)
    elif k == "T": c["T"] = int(v); c["common"]["horizon"] = int(v)
    elif k == "lean": c["knobs"]["lean"] = v == "1"
st = torch.from_numpy(c["states"]).cuda(); torch.cuda.synchronize()
with D.make(c, **c["knobs"]) as pl:
    pl.solve_n_async_device(16, st.data_ptr()); pl.sync()
    r1 = pl.recovery_count()
print(" ".join(sys.argv[1:]), "S =", c["B"] * ((c["K"] + 63) // 64 + 1), "first batch recoveries", r1, flush=True)
