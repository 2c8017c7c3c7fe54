"""Differential sweep of BN_FLAG_UNORDERED_OUTPUTS: the same host loop (forward_state_async + first_action per step) on a host-paced handle
with unordered outputs and on a plain one, with random calls in between -- order_outputs + a copy on the stream, getters, setters, batches,
pauses long enough for the age guard -- every copy, action and final buffer compared bit for bit.  usage: fuzz_unordered.py [cases] [seed]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
from benchnav_amd import NativeMPPI, synth   # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(cases):
    K = int(rng.choice([128, 256, 512, 1024])); T = int(rng.integers(5, 51)); G = int(rng.choice([64, 128, 256]))
    ref = bool(rng.integers(0, 2)); lean = bool(rng.integers(0, 4) == 0)
    inst = synth.make_instance(G, seed=int(rng.integers(0, 100)))
    n = int(rng.integers(6, 30))
    st = inst.start.numpy().astype(np.float32).copy()
    states, plan = [], []
    for i in range(n):
        if rng.integers(0, 12) == 0:
            st = st + np.array([4.0, -3.0, 0.7], np.float32)
        states.append(st.copy())
        st = st + np.array([0.08, 0.05, 0.02], np.float32) * rng.uniform(0.3, 1.6, 3).astype(np.float32)
        plan.append(int(rng.integers(0, 9)))
    n_out = T * 2 + (T + 1) * 3
    st_dev = torch.from_numpy(states[0]).cuda()
    res = {}
    for paced in (True, False):
        with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, seed=case, stream=0, host_paced=paced, unordered_outputs=paced,
                        reference_order=ref, lean=lean) as pl:
            pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
            rec = []
            for i, s in enumerate(states):
                o = torch.full((n_out,), float("nan"), device="cuda")
                torch.cuda.current_stream().synchronize()
                pl.forward_state_async(s, None, 0, o.data_ptr())
                rec.append(pl.first_action().copy())
                ev = plan[i]
                if ev == 0:
                    pl.order_outputs(); rec.append(o.clone())
                elif ev == 1:
                    rec.append(pl.weights()); rec.append(o.clone())
                elif ev == 2:
                    rec.append(pl.get_mean())
                elif ev == 3:
                    pl.solve_n_async_device(3, st_dev.data_ptr())
                elif ev == 4:
                    pl.set_goal(inst.goal.numpy() - 0.5 * (i % 3))
                elif ev == 5:
                    pl.order_outputs(); pl.order_outputs(); rec.append(o[: 2 * T].clone())
                elif ev == 6 and i % 5 == 0:
                    time.sleep(0.025)
                rec.append(o)
            pl.flush(); torch.cuda.synchronize(); pl.sync()
            rec += [pl.costs(), pl.weights(), pl.get_mean()]
            res[paced] = [r.cpu().numpy() if torch.is_tensor(r) else np.asarray(r) for r in rec]
    ok = len(res[True]) == len(res[False]) and all(np.array_equal(a, b) for a, b in zip(res[True], res[False]))
    if not ok:
        bad += 1
        print(f"case {case}: K={K} T={T} G={G} ref={ref} lean={lean} plan={plan} DIFFERS", flush=True)
print(f"{cases} cases, {bad} differ")
sys.exit(1 if bad else 0)
