"""A11: the duck-typed boundary, pinned against the REAL reference classes (CPU).

tests/golden/make_golden.py runs benchnav_amd's extractors -- `_planner_inputs` (MPPI, DWA) and `env_inputs` (the batched
environment) -- on the reference's own UnicycleModel / Objectives / GridMap / PlanetaryEnv and stores what they return
(boundary.npz) and which attributes the real objects carry (boundary.json).  Here reference-SHAPED objects are rebuilt from
nothing but those recorded attribute names and the stored arrays, and the extractors must give the same answers: a private
attribute renamed in benchnav_amd breaks this test, one renamed in the reference breaks the regeneration.
"""
import json
import os
import types

import numpy as np
import pytest
import torch

from benchnav_amd.env import env_inputs
from benchnav_amd.mppi import REFERENCE_READS, _planner_inputs

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def pinned():
    with open(os.path.join(HERE, "boundary.json")) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(HERE, "boundary.npz"))


def _shaped(meta, cls, **values):
    """An object with EXACTLY the attribute names the real `cls` instance had; the ones a value is given for hold it, the rest
    hold a sentinel that fails loudly when anything is done with it."""
    class _Unset:
        def __getattr__(self, k):
            raise AssertionError(f"{cls}: an attribute the fixture holds no value for was used")
    names = meta["attributes"][cls]
    unknown = set(values) - set(names)
    assert not unknown, f"{cls} has no attribute(s) {sorted(unknown)} in the reference"
    obj = types.SimpleNamespace(**{k: _Unset() for k in names})
    for k, v in values.items():
        setattr(obj, k, v)
    return obj


def _rebuild(meta, fx, mode):
    sc = fx["inf_scalars"]
    G, res = int(sc[0]), float(sc[1])
    latent = torch.distributions.Normal(torch.as_tensor(fx["MU"]), torch.as_tensor(fx["SG"]))
    assert meta["latent_distribution"] == "Normal" and "latent_models" in meta["distributions"]
    gm = _shaped(meta, "GridMap", grid_size=G, resolution=res, x_limits=(float(sc[2]), float(sc[3])), y_limits=(float(sc[4]), float(sc[5])),
                 distributions={"latent_models": latent, "predictions": latent})
    cfg = _shaped(meta, "ModelConfig", mode=mode)
    tm = _shaped(meta, "TraversabilityModel", _risks=torch.as_tensor(fx["inf_risks"]) if mode == "inference" else None)
    dyn = _shaped(meta, "UnicycleModel", _grid_map=gm, _model_config=cfg, _traversability_model=tm,
                  min_action=torch.as_tensor(fx["u_min"]), max_action=torch.as_tensor(fx["u_max"]))
    obj = _shaped(meta, "Objectives", _goal_pos=torch.as_tensor(fx["inf_goal"]), _stuck_threshold=float(sc[6]))
    return gm, dyn, obj


def test_every_attribute_the_product_reads_exists_on_the_real_classes(pinned):
    meta, _ = pinned
    for cls, reads in REFERENCE_READS.items():
        have = meta["attributes"][cls]
        assert all(r in have for r in reads), (cls, [r for r in reads if r not in have])
    # and the table is what the extractors really touch: objects carrying ONLY these names are enough (next tests)


def test_planner_inputs_on_reference_shaped_objects_match_the_real_ones(pinned):
    meta, fx = pinned
    _, dyn, obj = _rebuild(meta, fx, "inference")
    a = _planner_inputs(dyn, obj)
    assert torch.equal(torch.as_tensor(a["risks"]), torch.as_tensor(fx["inf_risks"])) and a["slip_std"] is None
    got = np.array([a["grid_size"], a["resolution"], *a["x_limits"], *a["y_limits"], a["stuck_threshold"]], np.float64)
    assert np.array_equal(got, fx["inf_scalars"])
    assert a["goal"].dtype == torch.int64 and np.array_equal(a["goal"].numpy(), fx["inf_goal"])      # test_mppi.py:133 passes an int64 goal
    assert isinstance(a["grid_size"], int) and isinstance(a["resolution"], float)


def test_sampled_slip_inputs_come_from_the_latent_model(pinned):
    meta, fx = pinned
    _, dyn, obj = _rebuild(meta, fx, "observation")
    b = _planner_inputs(dyn, obj, sampled_slip=True)
    assert np.array_equal(b["risks"].numpy(), fx["obs_risks"]) and np.array_equal(b["slip_std"].numpy(), fx["obs_slip_std"])
    got = np.array([b["grid_size"], b["resolution"], *b["x_limits"], *b["y_limits"], b["stuck_threshold"]], np.float64)
    assert np.array_equal(got, fx["obs_scalars"])


def test_mode_mismatches_raise_what_the_reference_raises(pinned):
    meta, fx = pinned
    # the reference's own MPPI fails with TypeError on observation-mode dynamics (recorded from the real class) -- so does the extractor
    assert meta["raised"]["reference_mppi_on_observation_mode"] == "TypeError"
    assert meta["raised"]["observation_without_sampled_slip"] == "TypeError" and meta["raised"]["inference_with_sampled_slip"] == "TypeError"
    _, dyn_o, obj_o = _rebuild(meta, fx, "observation")
    with pytest.raises(TypeError):
        _planner_inputs(dyn_o, obj_o)
    _, dyn_i, obj_i = _rebuild(meta, fx, "inference")
    with pytest.raises(TypeError):
        _planner_inputs(dyn_i, obj_i, sampled_slip=True)


def test_env_inputs_on_a_reference_shaped_environment(pinned):
    meta, fx = pinned
    gm, _, _ = _rebuild(meta, fx, "observation")
    sc = fx["env_scalars"]
    env = _shaped(meta, "PlanetaryEnv", _grid_map=gm, _start_pos=torch.as_tensor(fx["env_start"]), _goal_pos=torch.as_tensor(fx["env_goal"]),
                  _delta_t=float(sc[0]), _time_limit=float(sc[1]), stuck_threshold=float(sc[2]), _goal_threshold=float(sc[3]), _seed=int(sc[4]))
    e = env_inputs(env)
    assert np.array_equal(e["latent_mean"].numpy(), fx["env_latent_mean"]) and np.array_equal(e["latent_std"].numpy(), fx["env_latent_std"])
    assert np.array_equal(e["start_pos"].numpy(), fx["env_start"]) and np.array_equal(e["goal_pos"].numpy(), fx["env_goal"])
    got = np.array([e["delta_t"], e["time_limit"], e["stuck_threshold"], e["goal_threshold"], e["seed"], e["grid_size"], e["resolution"],
                    *e["x_limits"], *e["y_limits"]], np.float64)
    assert np.array_equal(got, sc)
    # the initial heading the batched environment derives (planetary_env.py:128-141) is the real environment's
    d = fx["env_goal"] - fx["env_start"]
    assert abs(float(np.arctan2(d[1], d[0])) - float(fx["env_robot_state0"][2])) < 1e-6


def test_the_test_fakes_are_reference_shaped(pinned):
    """tests/helpers.py's stand-ins (used by the GPU class tests) carry no attribute the real classes lack."""
    meta, fx = pinned
    from helpers import FakeDynamics, FakeGridMap, FakeObjectives
    gm = FakeGridMap(8, 0.5, latent=(np.zeros((8, 8), np.float32), np.ones((8, 8), np.float32)))
    dyn = FakeDynamics(np.zeros((8, 8), np.float32), gm)
    obj = FakeObjectives(torch.tensor([1.0, 2.0]), 0.3)
    for fake, cls in ((gm, "GridMap"), (dyn, "UnicycleModel"), (obj, "Objectives"), (dyn._model_config, "ModelConfig"),
                      (dyn._traversability_model, "TraversabilityModel")):
        extra = set(vars(fake)) - set(meta["attributes"][cls])
        assert not extra, (cls, extra)
