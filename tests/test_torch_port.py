"""CPU: the PyTorch-CPU port used as bench.py's cpu_baseline against the reference fixtures.
Same ATen ops as the reference, so on the fixture's torch build the results are (near) bit-identical."""
import numpy as np
import pytest
import torch

from helpers import CASES, load_case, parity_metrics
from oracle import torch_port as TP


@pytest.mark.parametrize("name", CASES)
def test_port_matches_reference_fixture(name):
    fx = load_case(name)
    pb = TP.problem_from_fixture(fx)
    same_build = str(fx["torch_version"]) == torch.__version__ and str(fx["cpu_capability"]) == torch.backends.cpu.get_cpu_capability()
    for i in range(int(fx["n_solves"])):
        out = TP.solve(pb, torch.from_numpy(fx[f"state_{i}"]), torch.from_numpy(fx[f"mean_{i}"]), torch.from_numpy(fx[f"eps_{i}"]))
        got = {k: v.numpy() for k, v in out.items()}
        m = parity_metrics(got, fx, i)
        if same_build:
            assert m["X_max"] == 0.0 and m["U_max"] == 0.0, m            # identical kernels -> identical trajectories
            assert m["w_max"] <= 1e-6 and m["Ustar_max"] <= 1e-6 and m["cost_outliers"] == 0, m
        else:
            assert m["X_max"] <= 1e-4 and m["cost_outlier_frac"] <= 5e-3 and m["w_max"] <= 2e-2, m
