"""Map-instance I/O ("next" row N4): the reference's .pt layout (dataset_generator.py:302-305) <-> plain arrays."""
import numpy as np
import pytest
import torch
from torch.distributions import Normal

from benchnav_amd import io as bio
from benchnav_amd import synth


def _reference_style_file(path, G=32):
    """Exactly what DatasetGenerator writes: a dict of tensors and a dict of Normal distributions."""
    tensors = {"heights": torch.rand(G, G), "slopes": torch.rand(G, G) * 0.3, "t_classes": torch.randint(0, 10, (G, G)).float(),
               "colors": torch.rand(3, G, G)}
    dists = {"latent_models": Normal(synth.smooth_risk_map(G, 1) * 0.6, synth.slip_std_map(G, 1))}
    torch.save({"tensors": tensors, "distributions": dists}, path)
    return tensors, dists


def test_load_reference_layout_and_round_trip(tmp_path):
    p = str(tmp_path / "000_000.pt")
    tensors, dists = _reference_style_file(p)
    inst = bio.load_instance(p)
    assert inst.grid_size == 32 and inst.pred_mean is None
    for k in tensors:
        assert torch.equal(inst.tensors[k], tensors[k])
    assert torch.equal(inst.latent_mean, dists["latent_models"].mean) and torch.equal(inst.latent_std, dists["latent_models"].stddev)
    inst.pred_mean, inst.pred_std = inst.latent_mean * 1.1, inst.latent_std * 0.9      # e.g. TraversabilityPredictor output
    q = str(tmp_path / "copy.pt")
    bio.save_instance(q, inst)
    raw = torch.load(q, weights_only=False)                     # readable by the reference loader (test_mppi.py:42-44)
    assert set(raw) == {"tensors", "distributions"} and isinstance(raw["distributions"]["latent_models"], Normal)
    again = bio.load_instance(q)
    assert torch.equal(again.pred_mean, inst.pred_mean) and torch.equal(again.latent_std, inst.latent_std)


@pytest.mark.gpu
def test_instance_to_planner_and_closed_loop(tmp_path):
    from benchnav_amd import NativeMPPI
    p = str(tmp_path / "000_001.pt")
    _reference_style_file(p, G=64)
    inst = bio.load_instance(p)
    inputs = bio.planner_inputs(inst, "cvar", 0.9, num_samples=500, seed=2)
    assert inputs["risk"].shape == (64, 64) and (inputs["risk"].cpu() >= inst.latent_mean).all()
    with NativeMPPI(horizon=20, num_samples=256, grid_size=64, resolution=0.5) as pl:
        pl.set_map(inputs["risk"].cpu().numpy()); pl.set_goal([24.0, 24.0])
        pl.env_attach(inputs["latent_mean"].numpy(), inputs["latent_std"].numpy())
        states, rewards, done = pl.episode(50, [8.0, 8.0, 0.7])
    d0, d1 = np.linalg.norm(states[0, 0, :2] - [24, 24]), np.linalg.norm(states[-1, 0, :2] - [24, 24])
    assert np.isfinite(states).all() and d1 < d0                # the rover makes progress toward the goal
