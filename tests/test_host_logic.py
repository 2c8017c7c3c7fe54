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


def test_bench_device_mapping_and_census_object():
    """bench.py's rank -> device rule for the three launch shapes, and the parity_census object it prints (data from the fixtures)."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    assert bench.pick_device(3, 8, 8, False) == (3, False, False)            # every rank sees all GPUs
    assert bench.pick_device(3, 1, 8, False) == (0, False, True)             # every rank sees one: device 0, identities checked later
    assert bench.pick_device(0, 1, 8, False) == (0, False, True)             # ... by rank 0 as well: the check is a collective
    assert bench.pick_device(0, 1, 1, False) == (0, False, False)
    assert bench.pick_device(3, 1, 8, True) == (0, True, False)              # rehearsal on a shared GPU
    assert bench.pick_device(9, 8, 16, False)[0] is None and bench.pick_device(0, 0, 1, False)[0] is None
    assert bench.distinct_gpus(["a", "b"]) and not bench.distinct_gpus(["a", "a"])
    pc = bench.parity_census()
    c2, c5 = pc["configs[1] K=1024 T=50 G=256"], pc["configs[4] K=16384 T=100 G=512"]
    assert c2["default (spec)"]["rollouts"] >= 5e5 and 0 < c2["default (spec)"]["rate"] < 1.2e-4 and c2["BN_FLAG_REFERENCE_ORDER"]["rate"] < 3e-5
    assert 0 < c5["default (spec)"]["rate"] < 4e-4 and c5["BN_FLAG_REFERENCE_ORDER"]["rate"] < 3e-5
