"""Latency of the drop-in class used the reference way: one forward(state) per control step, outputs consumed on the host.
Per mode: host time of the forward() call itself (how long Python is busy before it returns), and the full step
forward() + read-back of action_seq[0] (what the reference loop does every control step, test_mppi.py:174-181)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
from helpers import load_case, mppi_for_fixture
torch.set_num_threads(1)
fx = load_case("c2")
import sys as _sys
quick = "--quick" in _sys.argv
combos = ((("philox", True, False, "cpu"), ("philox", True, False, "loop"), ("philox", True, False, "acts")) if quick else
          (("torch", True, False, "cuda"), ("torch_device", True, False, "cuda"), ("philox", True, False, "cuda"), ("philox", True, False, "cpu"),
           ("philox", False, False, "cuda"), ("philox", False, False, "cpu"), ("philox", False, True, "cpu"), ("philox", True, False, "loop"), ("philox", False, False, "loop")))
for mode, copy, lean, where in combos:
    solver = mppi_for_fixture(fx, noise=mode, copy_outputs=copy, store_controls=False, lean=lean, host_loop=({"loop": True, "acts": "actions"}.get(where, False)))      # "loop": MPPI(host_loop=True), CPU state
    state = torch.tensor(fx["state_0"], device="cpu" if where in ("loop", "acts") else where)      # "cpu": the host loop's state, taken by value (bn_mppi_forward_state_async)
    for _ in range(50): U, X = solver(state)
    solver.release(); torch.cuda.synchronize(); n = 500
    host = 0.0
    t = time.perf_counter()
    for _ in range(n):
        t0 = time.perf_counter()
        U, X = solver(state)
        host += time.perf_counter() - t0
        solver.order_outputs()              # (host_loop="actions": the read-back needs the stream ordered; a no-op otherwise)
        a = U[0].cpu()                      # the reference loop reads action_seq[0] every step
    dt = (time.perf_counter() - t) / n
    solver.release()
    # the same step with the first action taken from the tail's host mailbox (MPPI.first_action): no synchronisation, no copy
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        U, X = solver(state)
        a = solver.first_action()
    dm = (time.perf_counter() - t) / n
    solver.release()
    assert torch.equal(a, U[0].cpu())
    a = a.clone()
    # back-to-back forwards without a read-back: the rate the host can feed the GPU at
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): U, X = solver(state)
    solver.release(); torch.cuda.synchronize(); dq = (time.perf_counter() - t) / n
    print(f"noise={mode:12s} copy_outputs={copy!s:5s} lean={lean!s:5s} state={where:4s}: forward() host {host / n * 1e6:6.1f} us | forward()+readback {dt * 1e6:6.1f} us "
          f"({1 / dt:.0f} Hz) | forward()+first_action() {dm * 1e6:6.1f} us ({1 / dm:.0f} Hz) | back-to-back {dq * 1e6:6.1f} us", flush=True)
