"""bench.py --gpus N launches its own N ranks.  On CPU this runs the launcher in its
rehearsal mode -- spawn, rendezvous over gloo, gather, report -- which creates no planner and prints no `value`."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus_2_spawns_two_ranks_that_meet():
    r = _run(["--gpus", "2", "--rehearse"], {"BENCH_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                     # rank 0 alone prints the line
    out = json.loads(lines[0])
    assert out["rehearsal"] is True and "value" not in out
    assert out["n_gpus"] == 2 and out["world_size"] == 2 and out["backend"] == "gloo"
    assert sorted(w["rank"] for w in out["ranks"]) == [0, 1]
    assert len({w["pid"] for w in out["ranks"]}) == 2    # two processes, one per (would-be) GPU
    assert out["gathered_shape"] == [2, 2]
    assert out["slowest_per_repeat"] == [0.002, 0.004]   # max over ranks per repeat


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "2", "--rehearse"], {"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "must agree" in (r.stderr + r.stdout)


def test_single_rank_rehearsal_needs_no_launcher():
    r = _run(["--gpus", "1", "--rehearse"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["world_size"] == 1 and out["ranks"][0]["rank"] == 0


def test_more_gpus_than_visible_is_an_error_not_a_silent_single_rank():
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = _run(["--gpus", str(have + 2), "--steps", "5", "--warmup", "1"])
    assert r.returncode != 0
    assert "visible" in r.stderr
