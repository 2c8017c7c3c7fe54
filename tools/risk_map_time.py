import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from benchnav_amd.risk import infer_risk_map
from benchnav_amd import synth
G = 256
mean = (synth.smooth_risk_map(G, 1) * 0.7).cuda(); std = synth.slip_std_map(G, 1).cuda()
for metric in ("var", "cvar"):
    for n in (1000, 4000):
        for _ in range(3): infer_risk_map(mean, std, metric, 0.9, num_samples=n, seed=1)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): infer_risk_map(mean, std, metric, 0.9, num_samples=n, seed=1)
        torch.cuda.synchronize(); print(metric, n, f"{(time.perf_counter() - t) / 20 * 1e3:.3f} ms")
