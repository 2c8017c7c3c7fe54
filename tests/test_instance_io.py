"""Map-instance I/O ("next" row N4): the reference's .pt layout (dataset_generator.py:302-305) <-> plain arrays."""
import numpy as np
import pytest
import torch
from torch.distributions import Normal

from benchnav_amd import io as bio
from benchnav_amd import synth


def _reference_style_file(path, G=32):
    """Exactly what DatasetGenerator writes: a dict of tensors and a dict of Normal distributions."""
    tensors = {"heights": torch.rand(G, G), "slopes": torch.rand(G, G) * 0.3, "t_classes": torch.randint(0, 10, (G, G)),
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
    raw = torch.load(q, weights_only=False)                     # the reference's own call (test_mppi.py:42)
    assert set(raw) == {"tensors", "distributions"} and isinstance(raw["distributions"]["latent_models"], Normal)
    again = bio.load_instance(q)
    assert torch.equal(again.pred_mean, inst.pred_mean) and torch.equal(again.latent_std, inst.latent_std)


def test_file_the_reference_read_is_read_the_same_way():
    """tests/golden/instance_000_000.pt was written by benchnav_amd.io.save_instance and then loaded by the REFERENCE's own path
    (torch.load -> GridMap, test_mppi.py:42-44, grid_map.py:100-143) in make_golden.py; instance.npz is what the reference's
    objects held.  load_instance must hand out the same arrays with the same dtypes (t_classes stays int64)."""
    import os
    from helpers import GOLDEN_DIR, load_case
    fx = load_case("instance")
    inst = bio.load_instance(os.path.join(GOLDEN_DIR, "instance_000_000.pt"))
    assert inst.grid_size == int(fx["G"])
    for k in ("heights", "slopes", "t_classes", "colors"):
        got = inst.tensors[k].numpy()
        assert got.dtype == fx[f"tensor_{k}"].dtype and np.array_equal(got, fx[f"tensor_{k}"]), k
    assert inst.tensors["t_classes"].dtype == torch.int64
    for k in ("latent_mean", "latent_std", "pred_mean", "pred_std"):
        assert np.array_equal(getattr(inst, k).numpy(), fx[k]) and getattr(inst, k).dtype == torch.float32
    # the reference's lookups on its GridMap (grid_map.py:145-210) are plain [iy, ix] reads of these arrays
    G, res = int(fx["G"]), float(fx["res"])
    ij = np.clip(np.floor(fx["positions"][0, :, :2] / np.float32(res)), 0, G - 1).astype(int)
    for k in ("heights", "slopes", "t_classes"):
        assert np.array_equal(inst.tensors[k].numpy()[ij[:, 1], ij[:, 0]], fx[f"at_{k}"][0])
    assert np.array_equal(inst.latent_mean.numpy()[ij[:, 1], ij[:, 0]], fx["latent_mean_at"][0])


def test_untrusted_files_go_through_the_restricted_unpickler(tmp_path):
    import pickle

    class Evil:
        def __reduce__(self):
            return (print, ("arbitrary code ran",))
    p = str(tmp_path / "evil.pt")
    torch.save({"tensors": {}, "distributions": {"latent_models": Evil()}}, p)
    with pytest.raises(pickle.UnpicklingError):
        bio.load_instance(p)


@pytest.mark.gpu
def test_reference_cvar_map_of_the_instance_and_planning_on_it():
    """The hand-off end to end on the file the reference read: predictions -> risk-map kernel with the reference's own draw
    reproduces the CVaR map its UnicycleModel inferred; planning on that map equals the oracle on the same arrays."""
    import os
    from helpers import GOLDEN_DIR, load_case
    from benchnav_amd import NativeMPPI
    from benchnav_amd.risk import infer_risk_map
    from oracle import oracle as O
    fx = load_case("instance")
    inst = bio.load_instance(os.path.join(GOLDEN_DIR, "instance_000_000.pt"))
    G, res = int(fx["G"]), float(fx["res"])
    risk = infer_risk_map(inst.pred_mean, inst.pred_std, "cvar", 0.9, num_samples=fx["z"].shape[0], z=torch.from_numpy(fx["z"])).cpu().numpy()
    assert np.abs(risk - fx["risk_cvar"]).max() <= 5e-7           # VaR exact, CVaR up to the summation order (tests/test_risk_map.py)
    K, T = 128, 10
    eps = np.random.default_rng(0).standard_normal((K, T, 2)).astype(np.float32)
    state, goal = np.array([0.6, 0.7, 0.4], np.float32), np.array([3.2, 3.1], np.float32)
    with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=res, store_controls=True) as pl:
        pl.set_map(risk); pl.set_goal(goal)
        us, xs = pl.solve(state, eps)
        X, c = pl.states(), pl.costs()
    orc = O.solve(O.make_params(K, T, G, res, goal, trig=O.TRIG_SPEC), risk, state, np.zeros((T, 2), np.float32), eps)
    assert np.array_equal(X, orc["X"]) and np.array_equal(c, orc["cost"]) and np.abs(us[0] - orc["Ustar"]).max() <= 2e-6


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
