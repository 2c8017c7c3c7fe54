import sys, os, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import differential as D
seed = 2681
c = D.case(seed + 300_000); c["noise"] = "philox"
B, K, T = c["B"], c["K"], c["T"]
st = torch.from_numpy(c["states"]).cuda(); torch.cuda.synchronize()
cur = torch.cuda.current_stream().cuda_stream
def run(knobs, script, use_cur=True):
    with D.make(c, **({"stream": cur} if use_cur else {}), **knobs) as pl:
        for kind, m in script:
            if kind == "batch": pl.solve_n_async_device(m, st.data_ptr())
            else: pl.solve_async_device(st.data_ptr())
        return D.outputs(pl, c, knobs.get("lean", False)), pl.recovery_count()
script = [("batch", 16), ("single", 1), ("single", 1), ("batch", 5)]
plain_script = [("single", 1)] * 23
want, _ = run(dict(overlap=False), plain_script)
reps = int(sys.argv[1])
for name, knobs, scr, uc in (("plain again", dict(overlap=False), plain_script, True),
                         ("knobbed", c["knobs"], script, True),
                         ("knobbed own stream", c["knobs"], script, False),
                         ("knobbed no overlap", dict(c["knobs"], overlap=False), script, True),
                         ("knobbed full", dict(c["knobs"], lean=False), script, True),
                         ("knobbed batch23", c["knobs"], [("batch", 23)], True)):
    bad = {}; t0 = time.time()
    for i in range(reps):
        got, rec = run(knobs, scr, uc)
        d = [k for k, v in got.items() if not np.array_equal(v, want[k], equal_nan=True)]
        if d or rec: bad[i] = (",".join(d), rec)
    print(f"{name}: {len(bad)} bad of {reps} ({time.time() - t0:.1f}s)", list(bad.items())[:4], flush=True)
