import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import differential as D
for seed in (12, 33):
    c = D.case(seed + 300_000)
    print(seed, {k: c[k] for k in ("B", "K", "T", "G", "knobs")}, {k: c["common"][k] for k in ("resolution", "dt", "u_max", "reference_order", "shared_map", "lambda_")})
    # replay the script generation to print it
    rng = np.random.default_rng(93_000 + seed)
    B, K, T, G = c["B"], c["K"], c["T"], c["G"]
    _ = [rng.normal(0, 0.3, c["states"].shape) for _ in range(2)]; _ = rng.normal(0, 2.0, c["goals"].shape); _ = rng.standard_normal((T, 2))
    script = []
    for _ in range(int(rng.integers(2, 9))):
        kind = str(rng.choice(["batch", "batch", "batch", "single", "goal", "mean", "map", "top", "state", "expire", "expire", "sync"]))
        script.append((kind, int(rng.choice([2, 3, 4, 5, 7, 16, 20])), int(rng.integers(0, B)), int(rng.integers(0, 3))))
    print("  script", script)
    # minimal: run solves then compare top_samples between plain and knobbed
    st = torch.from_numpy(c["states"]).cuda(); torch.cuda.synchronize()
    res = []
    for knobs in (dict(overlap=False), c["knobs"]):
        with D.make(c, **knobs) as pl:
            for _ in range(3): pl.solve_async_device(st.data_ptr())
            b = 0
            s, w = pl.top_samples(min(5, K), b)
            ww = pl.weights(b); order = np.argsort(-ww, kind="stable")[:5]
            res.append((s, w, ww[order], order))
    (s0, w0, ww0, o0), (s1, w1, ww1, o1) = res
    print("  weights equal", np.array_equal(w0, w1), w0, w1, "top idx", o0, o1, "ties among top:", ww0)
    print("  states equal", np.array_equal(s0, s1), np.abs(s0 - s1).max())
