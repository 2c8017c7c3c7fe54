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
    # the launch-to-launch time (HIP event pair around the same K launches, two extra packets in a 0.2 ms window) against the
    # step time of the timed region: the same thing measured twice
    assert rf["kernel_ms"] <= 1.05 * out["ms_per_step"]
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["peak_measured_copy"] > 1000.0
    assert out["value"] > 20000 and out["unit"] == "solves/s" and out["host_cpu"]["logical_cpus"] >= 1
    assert out["sustained"]["value"] > 20000


def test_the_line_carries_the_reference_boundarys_own_figure():
    """`dropin_forward` / `value_dropin_forward`: benchnav_amd.MPPI.forward(state) + first_action() once per control step with the state
    living on the host (test/test_mppi.py:174-181) -- one launch per forward, the host-paced loop on, the mailbox value IS action_seq[0],
    and faster than the same loop without the opt-in."""
    out = _bench(["--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-batched"])
    d = out["dropin_forward"]
    assert out["value_dropin_forward"] == d["value"] and d["unit"] == "control steps/s"
    assert d["launches_per_forward"] == 1 and d["host_loop"] is True and d["first_action_equals_action_seq0"] is True
    assert d["value"] > 35000 and d["us_per_step"] < d["without_host_loop"]["us_per_step"] < d["with_cpu_readback"]["us_per_step"] + 10
    assert abs(sum(d["split_us"].values()) - d["us_per_step"]) < 0.5
    assert d["host_loop_actions"]["first_action_equals_action_seq0"] is True and d["host_loop_actions"]["us_per_step"] < d["without_host_loop"]["us_per_step"]


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
    # a multi-GPU run of the DEFAULT command carries the two BASELINE workloads that name 8 GPUs as objects of the same line
    for key, cfg in (("sharded_c4", "configs[3]"), ("sharded_c5", "configs[4]")):
        o = out[key]
        assert cfg in o["config"]["workload"] and o["scaling"] == "strong" and len(o["per_rank_seconds"]) == 2
        assert 0 < o["roofline"]["frac"] < 1 and o["value"] > 1000
    assert out["parity_census"]["configs[1] K=1024 T=50 G=256"]["BN_FLAG_REFERENCE_ORDER"]["beyond_1e-4"] == 0


def test_ranks_that_each_see_one_device_must_see_different_ones():
    """A launcher that gives every rank ONE visible device (LOCAL_RANK >= device_count): device 0 is taken and the ranks compare GPU
    identities once the group is up.  On this box both ranks see the same GPU: a clear refusal, not a silent double booking."""
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("needs a box where every rank sees exactly one device")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BENCH_SHARE_GPU")}
    env["BENCH_DIST_BACKEND"] = "gloo"
    port = 29000 + os.getpid() % 2000
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--no-extras"],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode != 0 and "do not see 2 different GPUs" in r.stderr, r.stderr[-2000:]


def test_a_single_rank_goes_through_rccl_when_asked():
    """BENCH_FORCE_DIST=1: init_process_group("nccl") with a world of one, barrier around the timed regions, the times gathered with
    a device all-gather, the names with all_gather_object -- the code an 8-GPU launch runs, executed on the one GPU of the box."""
    out = _bench(["--steps", "20", "--warmup", "5", "--no-extras", "--no-cpu-baseline"], {"BENCH_FORCE_DIST": "1"})
    assert out["collective_backend"] == "nccl" and out["n_gpus"] == 1 and out["value"] > 20000 and "rehearsal" not in out


@pytest.mark.parametrize("workload", ["c4", "c5"])
def test_the_multi_gpu_workloads_on_one_rank_over_rccl(workload):
    """--workload c4 (64 instances sharded over the ranks) and c5 (one K=16384 solve sharded by rollouts: ShardedMPPI, one
    all_gather_into_tensor of the partials per solve) with a world of one over RCCL."""
    out = _bench(["--workload", workload, "--steps", "10", "--warmup", "3"], {"BENCH_FORCE_DIST": "1"})
    assert out["collective_backend"] == "nccl" and out["scaling"] == "strong" and out["steps"] == 10
    assert ("configs[3]" if workload == "c4" else "configs[4]") in out["config"]["workload"]
    rf = out["roofline"]
    assert 0 < rf["frac"] < 1 and rf["kernel_ms"] > 0
    assert out["value"] > (500000 if workload == "c4" else 5000)


def test_the_multi_gpu_workloads_with_two_ranks_on_the_device():
    import torch
    share = {} if torch.cuda.device_count() >= 2 else {"BENCH_SHARE_GPU": "1", "BENCH_DIST_BACKEND": "gloo"}
    for workload in ("c4", "c5"):
        out = _bench(["--gpus", "2", "--workload", workload, "--steps", "6", "--warmup", "2"], share)
        assert out["n_gpus"] == 2 and out["world_size"] == 2 and len(out["per_rank_seconds"]) == 2 and out["value"] > 1000
