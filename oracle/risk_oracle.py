"""NumPy restatement of TraversabilityModel._infer_risk_map (TEST INFRASTRUCTURE ONLY).

Follows reference traversability_model.py:28-51 with the sampling, quantile and tail mean spelled out:
  samples = z * std + mean                      Normal.sample == normal_().mul_(std).add_(mean)  
  var     = torch.quantile(samples, q, dim=0)   'linear': rank = fp32(q) * (n-1); lerp(below, above, frac)   (:35)
  cvar    = nanmean(where(samples > var, samples, nan), dim=0)                                    (:38-42)
Pinned by tests/golden/riskmap.npz (outputs of the imported reference on the same z).
"""
import numpy as np

f32 = np.float32


def infer_risk_map(mean, std, metric, confidence=None, z=None):
    mean = np.asarray(mean, f32); std = np.asarray(std, f32)
    if metric == "expected_value":
        return mean.copy()
    z = np.asarray(z, f32)
    n = z.shape[0]
    samples = ((z * std[None]).astype(f32) + mean[None]).astype(f32)
    srt = np.sort(samples, axis=0)
    pos = f32(f32(confidence) * f32(n - 1))
    lo_f = np.floor(pos); lo = int(lo_f); hi = int(np.ceil(pos)); w = f32(pos - lo_f)
    below, above = srt[lo], srt[hi]
    d = (above - below).astype(f32)
    if abs(w) < 0.5:                                  # at::lerp
        var = (below + (w * d).astype(f32)).astype(f32)
    else:
        var = (above - (d * f32(f32(1) - w)).astype(f32)).astype(f32)
    if metric == "var":
        return var
    mask = samples > var[None]
    s = np.where(mask, samples, f32(0)).astype(np.float64).sum(axis=0)
    c = mask.sum(axis=0)
    with np.errstate(invalid="ignore", divide="ignore"):
        return (s / c).astype(f32)
