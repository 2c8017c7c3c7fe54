"""Shared test helpers: golden-fixture loading and the tiered parity metrics."""
from __future__ import annotations

import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["c1_basic", "c1_stuck", "c1_edge", "res03", "cvar", "var", "ragged", "c2", "c2_stuck"]


def load_case(name):
    z = np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"))
    return {k: z[k] for k in z.files}


def oracle_params_for(fx, trig):
    from oracle import oracle as O
    return O.make_params(int(fx["K"]), int(fx["T"]), int(fx["G"]), float(fx["res"]), fx["goal"],
                         thr=float(fx["thr"]), lambda_=float(fx["lam"]), sigma=fx["sigmas"].tolist(),
                         inv_var=fx["inv_var"].tolist(), u_min=fx["u_min"].tolist(),
                         u_max=fx["u_max"].tolist(), x_limits=tuple(fx["x_limits"].tolist()),
                         y_limits=tuple(fx["y_limits"].tolist()), trig=trig)


def parity_metrics(got, fx, i):
    """Tiered comparison of one solve against the reference fixture (SURVEY.md 8a parity spec).

    `got` maps U, X, cost, w, Ustar, Xstar to arrays (full K); fixture X/U may be strided.
    """
    s = int(fx["x_stride"])
    m = {}
    m["U_max"] = float(np.abs(got["U"][::s] - fx[f"U_{i}"]).max())
    m["X_max"] = float(np.abs(got["X"][::s] - fx[f"X_{i}"]).max())
    c_ref = fx[f"cost_{i}"]
    dc = np.abs(got["cost"] - c_ref)
    tol = 1e-3 * np.maximum(1.0, np.abs(c_ref))
    bad = dc > tol
    m["cost_outliers"] = int(bad.sum())
    m["cost_outlier_frac"] = float(bad.mean())
    # outliers must be cell / threshold flips: an integer multiple of 1e4 (+- small)
    resid = np.abs(dc[bad] - 1e4 * np.round(dc[bad] / 1e4)) if bad.any() else np.zeros(0)
    m["cost_outlier_resid"] = float(resid.max()) if resid.size else 0.0
    m["cost_max_inlier"] = float((dc[~bad] / np.maximum(1.0, np.abs(c_ref[~bad]))).max()) if (~bad).any() else 0.0
    m["w_max"] = float(np.abs(got["w"] - fx[f"w_{i}"]).max())
    m["Ustar_max"] = float(np.abs(got["Ustar"] - fx[f"Ustar_{i}"]).max())
    m["Ustar_rms"] = float(np.sqrt(np.mean((got["Ustar"] - fx[f"Ustar_{i}"]) ** 2)))
    m["Xstar_max"] = float(np.abs(got["Xstar"] - fx[f"Xstar_{i}"]).max())
    return m


# Tolerances against the reference (fp32; stated in DESIGN.md "Parity tiers").
TOL_REF = dict(U_max=0.0, X_max=1e-4, cost_outlier_frac=5e-3, cost_outlier_resid=0.5,
               w_max=5e-3, Ustar_max=2e-2, Ustar_rms=2e-3, Xstar_max=1e-3)


def assert_within(m, tol=TOL_REF, ctx=""):
    for k, v in tol.items():
        assert m[k] <= v, f"{ctx}: {k}={m[k]:.3e} exceeds {v:.1e} ({m})"
