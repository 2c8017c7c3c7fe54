import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
from helpers import FakeDynamics, FakeGridMap, FakeObjectives
from benchnav_amd import DWA, synth
G = 256
gm = FakeGridMap(G, 0.5); dyn = FakeDynamics(synth.smooth_risk_map(G, 0), gm); obj = FakeObjectives(torch.tensor([96.0, 96.0]), 0.3)
s = DWA(50, 3, 2, dyn, obj, torch.tensor([0.5, 1.0]), 0.1)
st = torch.tensor([32.0, 32.0, 0.7])
for _ in range(10): s(st)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(200): a, x = s(st)
torch.cuda.synchronize(); print(f"DWA.forward: {(time.perf_counter() - t) / 200 * 1e6:.0f} us")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(100): s(st)
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
