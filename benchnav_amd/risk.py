"""Risk-map precompute on the GPU: mirror of TraversabilityModel._infer_risk_map
(reference src/simulator/problem_formulation/traversability_model.py:28-51)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _capi

_METRICS = {"expected_value": _capi.BN_RISK_EXPECTED, "var": _capi.BN_RISK_VAR, "cvar": _capi.BN_RISK_CVAR}


def infer_risk_map(mean: torch.Tensor, std: torch.Tensor, inference_metric: str, confidence_value: Optional[float] = None,
                   num_samples: int = 1000, z: Optional[torch.Tensor] = None, seed: int = 0,
                   device: Optional[torch.device] = None) -> torch.Tensor:
    """(G,G) risk map from the predicted slip distribution Normal(mean, std).

    inference_metric / confidence_value follow ModelConfig (problem_formulation/utils.py:10-40).
    z: optional (num_samples, G, G) standard normals -- pass the reference's own draw
    (`torch.empty(n, G, G).normal_()`) to reproduce its map; None samples inside the kernel.
    Returns a float32 tensor on the GPU.  Raises without a GPU (no CPU fallback).
    """
    if inference_metric not in _METRICS:
        raise AssertionError(f"inference_metric must be one of {list(_METRICS)}")
    if inference_metric != "expected_value":
        assert confidence_value is not None and 0.0 <= confidence_value <= 1.0, \
            "confidence_value must be set between 0 and 1 when inference_metric is 'var' or 'cvar'."
    if not torch.cuda.is_available():
        raise RuntimeError("benchnav_amd.risk needs an MI355X (gfx950) device; there is no CPU fallback")
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    G = mean.shape[0]
    assert mean.shape == (G, G) and std.shape == (G, G)
    m = mean.detach().to(dev, torch.float32).contiguous()
    s = std.detach().to(dev, torch.float32).contiguous()
    out = torch.empty(G, G, device=dev, dtype=torch.float32)
    zd = None
    if z is not None:
        assert z.shape == (num_samples, G, G)
        zd = z.detach().to(dev, torch.float32).contiguous()
    lib = _capi.load()
    with torch.cuda.device(dev):
        rc = lib.bn_risk_map_infer(dev.index, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream),
                                   C.c_void_p(m.data_ptr()), C.c_void_p(s.data_ptr()), _capi.BN_MEM_DEVICE, G,
                                   _METRICS[inference_metric], float(confidence_value or 0.0), int(num_samples),
                                   C.c_void_p(zd.data_ptr() if zd is not None else None), _capi.BN_MEM_DEVICE, seed,
                                   C.c_void_p(out.data_ptr()), _capi.BN_MEM_DEVICE)
    if rc != _capi.BN_OK:
        raise _capi.BenchnavError(rc, lib.bn_risk_last_error().decode())
    return out
