import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
from helpers import load_case, mppi_for_fixture
fx = load_case("c2")
solver = mppi_for_fixture(fx, noise="philox", copy_outputs=True, store_controls=False)
state = torch.tensor(fx["state_0"], device="cuda")
for _ in range(50): solver(state)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(2000): U, X = solver(state)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host time per forward {1e6*(t1-t)/2000:.1f} us; incl. drain {1e6*(t2-t)/2000:.1f} us")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): solver(state)
pr.disable(); torch.cuda.synchronize(); pstats.Stats(pr).sort_stats("tottime").print_stats(10)
