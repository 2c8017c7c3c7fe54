"""Latency of the drop-in class used the reference way: one forward(state) per control step, outputs consumed on the host."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
from helpers import load_case, mppi_for_fixture
fx = load_case("c2")
for mode, copy in (("torch", True), ("philox", True), ("philox", False), ("torch_device", True)):
    solver = mppi_for_fixture(fx, noise=mode, copy_outputs=copy, store_controls=False)
    state = torch.tensor(fx["state_0"], device="cuda")
    for _ in range(20): U, X = solver(state)
    torch.cuda.synchronize(); t = time.perf_counter(); n = 300
    for _ in range(n):
        U, X = solver(state)
        a = U[0].cpu()                      # the reference loop reads action_seq[0] every step
    dt = (time.perf_counter() - t) / n
    print(f"noise={mode:12s} copy_outputs={copy}: {dt*1e6:7.1f} us per forward()+readback  ({1/dt:.0f} Hz)")
