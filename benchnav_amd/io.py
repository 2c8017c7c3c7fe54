"""Map-instance I/O and the prediction hand-off ("next" row N4).

BenchNav stores one map instance as `torch.save({"tensors": ..., "distributions": ...}, path)`
(reference src/data/dataset_generator.py:302-305; loaded at test/test_mppi.py:42-44):
  tensors        heights, slopes, t_classes (G,G) and colors (3,G,G)            grid_map.py:100-109
  distributions  latent_models: Normal(mean, std) per cell (the ground-truth slip model the environment
                 samples in observation mode) and optionally predictions: Normal (what
                 TraversabilityPredictor.predict returns, classifier_and_regressor.py `predict`)
The native planner wants plain float32 arrays; this module converts in both directions, so real
BenchNav instances (from the project's release archive) flow into the C ABI without reference code.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch
from torch.distributions import Normal


@dataclass
class MapInstance:
    grid_size: int
    tensors: Dict[str, torch.Tensor]                      # heights, slopes, t_classes, colors: CPU, dtypes as stored (t_classes is
                                                          # int64 in the reference's files, terrain_properties.py:430)
    latent_mean: torch.Tensor                             # (G,G)
    latent_std: torch.Tensor                              # (G,G)
    pred_mean: Optional[torch.Tensor] = None              # (G,G) predicted slip mean, if stored
    pred_std: Optional[torch.Tensor] = None
    extra: Dict[str, object] = field(default_factory=dict)


def _mean_std(dist) -> Tuple[torch.Tensor, torch.Tensor]:
    if isinstance(dist, (tuple, list)):
        m, s = dist
    else:                                                  # torch.distributions.Normal (or anything with mean/stddev)
        m, s = dist.mean, dist.stddev
    return m.detach().to("cpu", torch.float32).contiguous(), s.detach().to("cpu", torch.float32).contiguous()


def load_instance(path: str, trusted: bool = False) -> MapInstance:
    """Read a BenchNav instance file into plain CPU tensors (stored dtypes kept; the planner-facing arrays -- latent and
    predicted slip mean / std -- as float32).  The file pickles torch.distributions.Normal objects: by default it is read
    with torch's restricted unpickler and only that class allow-listed; trusted=True falls back to the full unpickler for
    files from a source you control (what the reference's own torch.load does, test_mppi.py:42)."""
    if trusted:
        item = torch.load(path, map_location="cpu", weights_only=False)
    else:
        with torch.serialization.safe_globals([Normal]):
            item = torch.load(path, map_location="cpu", weights_only=True)
    tensors = {k: v.detach().to("cpu").contiguous() for k, v in item["tensors"].items()}
    dists = item["distributions"]
    lat_m, lat_s = _mean_std(dists["latent_models"])
    G = lat_m.shape[-1]
    inst = MapInstance(grid_size=G, tensors=tensors, latent_mean=lat_m, latent_std=lat_s)
    if "predictions" in dists and dists["predictions"] is not None:
        inst.pred_mean, inst.pred_std = _mean_std(dists["predictions"])
    return inst


def save_instance(path: str, inst: MapInstance) -> None:
    """Write `inst` in the reference's on-disk format (dataset_generator.py:302-305)."""
    dists = {"latent_models": Normal(inst.latent_mean, inst.latent_std)}
    if inst.pred_mean is not None:
        dists["predictions"] = Normal(inst.pred_mean, inst.pred_std)
    torch.save({"tensors": dict(inst.tensors), "distributions": dists}, path)


def planner_inputs(inst: MapInstance, inference_metric: str = "cvar", confidence_value: Optional[float] = 0.9,
                   predictions=None, num_samples: int = 1000, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Everything the native planner and the device-side environment need from one instance:
    risk map (via the GPU risk-map kernel, = UnicycleModel.__init__ -> _infer_risk_map) and the latent model.
    `predictions` = the Normal returned by TraversabilityPredictor.predict (or a (mean, std) pair); defaults to
    the predictions stored in the file, else to the latent model itself (an oracle predictor)."""
    from .risk import infer_risk_map
    if predictions is not None:
        pm, ps = _mean_std(predictions)
    elif inst.pred_mean is not None:
        pm, ps = inst.pred_mean, inst.pred_std
    else:
        pm, ps = inst.latent_mean, inst.latent_std
    risk = infer_risk_map(pm, ps, inference_metric, confidence_value, num_samples=num_samples, seed=seed)
    return {"risk": risk, "latent_mean": inst.latent_mean, "latent_std": inst.latent_std}
