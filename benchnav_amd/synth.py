"""Seeded synthetic inputs for tests and benchmarks.

Pure torch-CPU formulas; no reference code involved.  Geometry follows the
reference GridMap convention (grid_map.py:42-50): origin (0, 0), limits
[0, G*res], risk map indexed [iy, ix].
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F


def smooth_risk_map(grid_size: int, seed: int = 0, coarse: int = 16, peak: float = 0.95) -> torch.Tensor:
    """Bicubic-upsampled uniform noise, min-max scaled to [0, peak] (float32 [G, G])."""
    g = torch.Generator().manual_seed(seed)
    n = max(2, grid_size // coarse)
    base = torch.rand(1, 1, n, n, generator=g)
    up = F.interpolate(base, size=(grid_size, grid_size), mode="bicubic", align_corners=False)[0, 0]
    up = (up - up.min()) / (up.max() - up.min())
    return (up * peak).to(torch.float32).contiguous()


def iid_risk_map(grid_size: int, seed: int = 0, peak: float = 0.95) -> torch.Tensor:
    """Cell-wise independent uniform risk in [0, peak): worst case for cache/LDS locality."""
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(grid_size, grid_size, generator=g) * peak).to(torch.float32)


def slip_std_map(grid_size: int, seed: int = 0, lo: float = 0.05, hi: float = 0.2) -> torch.Tensor:
    """Heteroscedastic slip std map for sampled-slip (config 3) workloads."""
    g = torch.Generator().manual_seed(seed + 7919)
    return (lo + (hi - lo) * torch.rand(grid_size, grid_size, generator=g)).to(torch.float32)


@dataclass
class Instance:
    """One planning instance: map + start/goal."""
    risk: torch.Tensor          # (G, G) float32
    start: torch.Tensor         # (3,)  x, y, theta
    goal: torch.Tensor          # (2,)
    grid_size: int
    resolution: float


def make_instance(grid_size: int, seed: int = 0, resolution: float = 0.5, kind: str = "smooth",
                  jitter: bool = False) -> Instance:
    """Start at (0.25, 0.25)*extent heading pi/4, goal at (0.75, 0.75)*extent.

    With jitter=True the start/goal are perturbed per seed (config 4's independent
    start-goal instances) by up to 5 % of the extent.
    """
    risk = smooth_risk_map(grid_size, seed) if kind == "smooth" else iid_risk_map(grid_size, seed)
    ext = grid_size * resolution
    start = torch.tensor([0.25 * ext, 0.25 * ext, math.pi / 4], dtype=torch.float32)
    goal = torch.tensor([0.75 * ext, 0.75 * ext], dtype=torch.float32)
    if jitter:
        g = torch.Generator().manual_seed(10_000 + seed)
        d = (torch.rand(4, generator=g) - 0.5) * 0.1 * ext
        start[:2] += d[:2]
        goal += d[2:]
    start[:2] = snap_to_free(risk, start[:2], resolution)
    goal = snap_to_free(risk, goal, resolution)
    return Instance(risk, start, goal, grid_size, resolution)


def snap_to_free(risk: torch.Tensor, pos: torch.Tensor, resolution: float, max_risk: float = 0.35) -> torch.Tensor:
    """Move `pos` to the centre of the nearest cell whose risk is below `max_risk`.

    Keeps benchmark instances meaningful: a start inside a stuck region makes every
    rollout collide at every step (costs ~5e5, weights decided by the last fp32 ulp).
    """
    G = risk.shape[0]
    iy, ix = torch.meshgrid(torch.arange(G), torch.arange(G), indexing="ij")
    cx = (ix.float() + 0.5) * resolution
    cy = (iy.float() + 0.5) * resolution
    d2 = (cx - pos[0]) ** 2 + (cy - pos[1]) ** 2
    d2 = torch.where(risk < max_risk, d2, torch.full_like(d2, float("inf")))
    j = int(torch.argmin(d2))
    if not math.isfinite(float(d2.view(-1)[j])):
        return pos.clone()
    return torch.stack([cx.view(-1)[j], cy.view(-1)[j]]).to(torch.float32)


def torch_cpu_noise(seed: int, K: int, T: int, n_solves: int) -> torch.Tensor:
    """The reference's noise stream on torch's CPU generator.

    mppi.py:55 seeds the global generator, mppi.py:105-107 discards one (K,T,2)
    draw in the constructor, then every forward draws one (K,T,2) block
    (mppi.py:149-151).  Bitwise equality with the reference holds only on the same
    torch build + CPU dispatch level, so golden fixtures store their noise.
    """
    g = torch.Generator().manual_seed(seed)
    torch.empty(K, T, 2).normal_(generator=g)
    return torch.stack([torch.empty(K, T, 2).normal_(generator=g) for _ in range(n_solves)])
