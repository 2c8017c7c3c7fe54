import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from helpers import native_outputs, oracle_metrics
from oracle import oracle as O
from benchnav_amd import NativeMPPI, synth
for K, T in [(300, 33), (300, 34), (64, 33), (320, 33), (300, 31), (300, 35), (300, 37)]:
    G = 256
    inst = synth.make_instance(G, seed=21)
    rng = np.random.default_rng(1)
    eps = rng.standard_normal((K, T, 2)).astype(np.float32)
    mean = np.clip(rng.standard_normal((T, 2)) * 0.2 + [0.6, 0.0], [0, -1], [1, 1]).astype(np.float32)
    p = O.make_params(K, T, G, 0.5, inst.goal.numpy(), trig=O.TRIG_SPEC)
    orc = O.solve(p, inst.risk.numpy(), inst.start.numpy(), mean, eps)
    for pipe in (True, False):
        with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, store_controls=True, pipeline=pipe) as pl:
            pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy()); pl.set_mean(mean)
            us, xs = pl.solve(inst.start.numpy(), eps)
            m = oracle_metrics(native_outputs(pl, us, xs), orc)
            d = np.abs(us[0] - orc["Ustar"])
            print(K, T, pipe, m["Ustar_max"], np.argwhere(d > 1e-4).tolist()[:6])
