"""GPU: bench.py end to end on the box -- the single-rank line carries what the contract asks for, and the self-launched
two-rank path runs on hardware (two ranks sharing the one GPU of the test box over gloo: a rehearsal, flagged as such)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_single_rank_line():
    out = _bench(["--steps", "20", "--warmup", "5", "--no-extras", "--no-cpu-baseline"])
    assert out["n_gpus"] == 1 and out["world_size"] == 1 and out["steps"] == 20 and out["repeats"] == 100
    assert out["ms_per_step_min"] <= out["ms_per_step"] <= out["ms_per_step_max"]
    rf = out["roofline"]
    assert rf["kernel_ms"] <= out["ms_per_step"]            # a kernel cannot take longer than the step it lives in
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["peak_measured_copy"] > 1000.0
    assert out["value"] > 20000 and out["unit"] == "solves/s" and out["host_cpu"]["logical_cpus"] >= 1
    assert out["sustained"]["value"] > 20000


def test_gpus_2_self_launches_two_ranks_on_the_device():
    import torch
    share = {} if torch.cuda.device_count() >= 2 else {"BENCH_SHARE_GPU": "1", "BENCH_DIST_BACKEND": "gloo"}
    out = _bench(["--gpus", "2", "--steps", "50", "--warmup", "10"], share)
    assert out["n_gpus"] == 2 and out["world_size"] == 2 and len(out["devices"]) == 2
    assert out["per_rank_solves"] == [50, 50] and len(out["per_rank_seconds"]) == 2
    assert out["scaling"] == "weak" and "cpu_baseline" not in out      # the CPU baseline is an N = 1 leg
    if share:
        assert "rehearsal" in out
    assert out["value"] > 10000
