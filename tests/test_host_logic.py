"""CPU: host-side logic (synthetic inputs, noise stream, reference-facing argument checks)."""
import math

import numpy as np
import pytest
import torch

from benchnav_amd import synth


def test_smooth_map_range_and_determinism():
    a, b = synth.smooth_risk_map(64, 3), synth.smooth_risk_map(64, 3)
    assert torch.equal(a, b) and a.shape == (64, 64) and a.dtype == torch.float32
    assert float(a.min()) == 0.0 and abs(float(a.max()) - 0.95) < 1e-6
    assert not torch.equal(a, synth.smooth_risk_map(64, 4))


def test_instance_start_and_goal_are_on_free_cells():
    for seed in range(4):
        inst = synth.make_instance(128, seed=seed, jitter=True)
        for pos in (inst.start[:2], inst.goal):
            ix, iy = int(pos[0] / inst.resolution), int(pos[1] / inst.resolution)
            assert float(inst.risk[iy, ix]) < 0.35
    assert abs(float(synth.make_instance(64).start[2]) - math.pi / 4) < 1e-6


def test_noise_stream_matches_the_reference_recipe():
    """manual_seed; one discarded (K,T,2) draw; then one (K,T,2) draw per solve."""
    K, T = 48, 7
    torch.manual_seed(42)
    torch.empty(K, T, 2).normal_()
    want = [torch.empty(K, T, 2).normal_() for _ in range(3)]
    got = synth.torch_cpu_noise(42, K, T, 3)
    for a, b in zip(want, got):
        assert torch.equal(a, b)


def test_fixture_noise_is_the_torch_cpu_stream_when_the_dispatch_level_matches():
    from helpers import load_case
    fx = load_case("c1_basic")
    if str(fx["cpu_capability"]) != torch.backends.cpu.get_cpu_capability() or str(fx["torch_version"]) != torch.__version__:
        pytest.skip("different torch build / CPU dispatch level: normal_() streams are not comparable")
    eps = synth.torch_cpu_noise(int(fx["seed"]), int(fx["K"]), int(fx["T"]), int(fx["n_solves"]))
    for i in range(int(fx["n_solves"])):
        assert np.array_equal(eps[i].numpy(), fx[f"eps_{i}"])


def test_mppi_constructor_asserts_like_the_reference():
    """Shape asserts come first (mppi.py:58-66), before any device is touched."""
    from benchnav_amd.mppi import MPPI

    class Dyn:
        min_action = torch.tensor([0.0, -1.0, 0.0]); max_action = torch.tensor([1.0, 1.0])
    with pytest.raises(AssertionError, match="minimum actions"):
        MPPI(5, 8, 3, 2, Dyn(), object(), torch.tensor([0.5, 0.5]), 0.5)
    Dyn.min_action = torch.tensor([0.0, -1.0])
    with pytest.raises(AssertionError, match="sigmas"):
        MPPI(5, 8, 3, 2, Dyn(), object(), torch.tensor([0.5]), 0.5)
