"""Closed-loop control rate on the device: B instances, n steps (one launch per step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from benchnav_amd import NativeMPPI, synth
G, K, T = 256, 1024, 50
for B in (1, 8, 64):
    insts = [synth.make_instance(G, seed=s, jitter=True) for s in range(min(B, 8))]
    pl = NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, num_instances=B, stream=0)
    for b in range(B):
        it = insts[b % len(insts)]; pl.set_map(it.risk.numpy(), b); pl.set_goal(it.goal.numpy(), b)
    lat = np.stack([insts[b % len(insts)].risk.numpy() for b in range(B)]); std = np.full_like(lat, 0.05)
    pl.env_attach(lat, std)
    starts = np.stack([insts[b % len(insts)].start.numpy() for b in range(B)])
    pl.episode(200, starts)
    n = 2000
    t = time.perf_counter(); states, rewards, done = pl.episode(n, starts); dt = time.perf_counter() - t
    dist0 = np.linalg.norm(states[0, :, :2] - np.stack([insts[b % len(insts)].goal.numpy() for b in range(B)]), axis=1)
    dist1 = np.linalg.norm(states[-1, :, :2] - np.stack([insts[b % len(insts)].goal.numpy() for b in range(B)]), axis=1)
    print(f"B={B}: {n} closed-loop steps in {dt*1e3:.1f} ms -> {dt/n*1e6:.1f} us/step, {B*n/dt:.0f} control steps/s; reached {int((done>=0).sum())}/{B}; mean distance to goal {dist0.mean():.1f} -> {dist1.mean():.1f} m; mean reward {rewards.mean():.2f}")
    pl.close()
