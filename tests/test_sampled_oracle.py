"""CPU: the oracle's sampled-slip solve (BASELINE config 3) pinned against the `sampled` golden fixture -- the
reference's observation-mode components (traversability_model.py:65-69, robot_model.py:59-100, objectives.py:29-65)
driven through the loop of mppi.py:150-214 with every Normal draw captured (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from helpers import TOL_REF, assert_within, load_case, oracle_params_for, parity_metrics


def _as_solve_fixture(fx):
    d = dict(fx)
    d["x_stride"] = 1
    for k in ("U", "X", "cost", "w", "Ustar", "Xstar"):
        d[f"{k}_0"] = fx[k]
    return d


@pytest.mark.parametrize("trig", ["libm", "spec"])
def test_sampled_oracle_matches_reference_components(trig):
    from oracle import oracle as O
    fx = load_case("sampled")
    p = oracle_params_for(fx, O.TRIG_LIBM if trig == "libm" else O.TRIG_SPEC)
    got = O.solve_sampled(p, fx["MU"], fx["SG"], fx["state"], fx["mean"], fx["eps"], fx["zt"], fx["zc"], fx["zo"])
    m = parity_metrics(got, _as_solve_fixture(fx), 0)
    assert_within(m, TOL_REF, ctx=f"sampled/{trig}")
    assert m["X_max"] < 2e-5 and m["cost_outliers"] <= 1, m


def test_sampled_oracle_with_zero_std_is_the_inference_solve():
    from oracle import oracle as O
    fx = load_case("sampled")
    p = oracle_params_for(fx, O.TRIG_SPEC)
    a = O.solve_sampled(p, fx["MU"], np.zeros_like(fx["SG"]), fx["state"], fx["mean"], fx["eps"], fx["zt"], fx["zc"], fx["zo"])
    b = O.solve(p, fx["MU"], fx["state"], fx["mean"], fx["eps"])
    for k in ("U", "X", "cost", "w", "Ustar", "Xstar"):
        assert np.array_equal(a[k], b[k]), k


def test_sampled_draws_change_the_rollouts():
    from oracle import oracle as O
    fx = load_case("sampled")
    p = oracle_params_for(fx, O.TRIG_SPEC)
    a = O.solve_sampled(p, fx["MU"], fx["SG"], fx["state"], fx["mean"], fx["eps"], fx["zt"], fx["zc"], fx["zo"])
    b = O.solve(p, fx["MU"], fx["state"], fx["mean"], fx["eps"])
    assert np.abs(a["X"] - b["X"]).max() > 1e-3
    # transit and cost draws are independent streams: swapping the cost draws leaves the trajectories alone
    c = O.solve_sampled(p, fx["MU"], fx["SG"], fx["state"], fx["mean"], fx["eps"], fx["zt"], -fx["zc"], fx["zo"])
    assert np.array_equal(a["X"], c["X"]) and not np.array_equal(a["cost"], c["cost"])
