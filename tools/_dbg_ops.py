import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import differential as D
seed = int(sys.argv[1])
c = D.case(seed + 300_000)
print(seed, {k: c[k] for k in ("B", "K", "T", "G", "knobs")}, {k: c["common"][k] for k in ("resolution", "dt", "u_max", "reference_order", "shared_map", "lambda_")})
rng = np.random.default_rng(93_000 + seed)
B, K, T, G = c["B"], c["K"], c["T"], c["G"]
_ = [rng.normal(0, 0.3, c["states"].shape) for _ in range(2)]; _ = rng.normal(0, 2.0, c["goals"].shape); _ = rng.standard_normal((T, 2))
script = []
for _ in range(int(rng.integers(2, 9))):
    kind = str(rng.choice(["batch", "batch", "batch", "single", "goal", "mean", "map", "top", "state", "expire", "expire", "sync"]))
    script.append((kind, int(rng.choice([2, 3, 4, 5, 7, 16, 20])), int(rng.integers(0, B)), int(rng.integers(0, 3))))
script.append(("batch", int(rng.choice([3, 5, 17])), 0, 0))
print("script", script)
cnt = {}
for i in range(int(sys.argv[2])):
    r = D.ops(seed); cnt[r] = cnt.get(r, 0) + 1
print(cnt)
