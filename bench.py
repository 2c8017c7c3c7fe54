#!/usr/bin/env python3
"""MPPI solve-steps/sec on MI355X (BASELINE.json metric) -- one JSON line on stdout.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 and no WORLD_SIZE in the environment launches the N ranks itself (one process per GPU under
torch.distributed.run, RCCL = backend "nccl"); under an external launcher WORLD_SIZE must equal N.  Fewer than N visible
GPUs is an error.

Workload (config.workload): BASELINE.json configs[1] -- a single 256x256 map, K=1024 rollouts, T=50 steps, one planning
instance per GPU.  A "step" is one complete MPPI solve (noise sampling, K x T rollout with per-step map lookups,
stage/terminal/control costs, softmin weights, weighted control reduction, optimal-sequence rollout, warm-start update)
-- the work of the reference's MPPI.forward (mppi.py:130-219).  Successive steps are warm-started from the previous U*,
so they form a dependent chain exactly like the reference's closed loop; the planner state is held fixed (open-loop
variant).  Inputs (map, goal, state, mean) are resident in HBM before the timed region.  With N > 1 every
rank plans its own map seed (instance sharding, no data-path collective); RCCL carries only the barrier around the timed
region and the final gather of the per-rank times.

Timing: exactly K steps bracketed by barrier + synchronize, max over ranks -- repeated (`repeats`) so that a K of 20 is not
a 0.3 ms sample; `ms_per_step` / `value` are the MEDIAN repeat, the min is reported beside it.  Per rank and repeat: barrier,
synchronize, clock, the K steps, synchronize, clock, barrier.  The closing barrier sits BEHIND the second clock reading: the
job time of a repeat is the maximum over the ranks of intervals that start together, which is what a barrier in front of the
clock would yield -- minus the collective's own latency (an RCCL barrier is ~130 us here, more than half of a 20-step region).

Extra objects on the line:
  roofline      dominant kernel (rollout) against the HBM roofline: algorithmic bytes per launch over the kernel's mean
                launch-to-launch duration, from a HIP event pair on the launch stream around the same K launches on the same
                handle right after the timed repeats (the timed region itself holds nothing but the solves)
  no_overlap    the headline with every launch on one stream: the configuration whose per-kernel duration rocprofv3 reports
                directly (by default consecutive dependent solves overlap on two streams, bn_mppi_solve_n_async)
  cpu_baseline  the PyTorch-CPU port of the reference (oracle/torch_port.py) timed on this host (N = 1 only)
  batched       64 instances per launch (config 4's per-node batch on one GPU): the regime where the HBM roofline is
                meaningful; with its lean-mode line (no trajectory dump) beside the full-API one
  dropin_forward / value_dropin_forward   the reference's own boundary (test/test_mppi.py:174-181): benchnav_amd.MPPI.forward(state) +
                first_action() once per control step, the state living on the host and changing every step -- with the time split
                (forward call, first-action wait, host environment step), the same loop without `host_loop` and with `U[0].cpu()`
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

G, K, T, RES = 256, 1024, 50, 0.5
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E datasheet peak (MI355X_MICROARCH.md)
MIN_TIMED_STEPS = 2000         # repeats = ceil(MIN_TIMED_STEPS / steps)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--noise", choices=["philox", "injected"], default="philox",
                    help="philox: sampled inside the rollout kernel (timed); injected: pre-generated eps resident in HBM")
    ap.add_argument("--batched-instances", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batched", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline + roofline only")
    ap.add_argument("--no-overlap", action="store_true",
                    help="keep every launch on one stream (BN_FLAG_NO_OVERLAP): the configuration whose per-kernel duration rocprofv3 "
                         "reports directly; by default consecutive dependent solves overlap on two streams")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--workload", choices=["c2", "c4", "c5"], default="c2",
                    help="c2 (default, the headline): BASELINE configs[1], one K=1024 instance per GPU.  c4: configs[3], 64 independent "
                         "instances sharded over the ranks (64/N per GPU, one launch per step).  c5: configs[4], ONE K=16384, T=100, "
                         "512x512 solve sharded by rollouts over the ranks (RCCL all-gather of the softmin partials per solve)")
    ap.add_argument("--rehearse", action="store_true",
                    help="launcher check without a GPU: spawn the ranks, rendezvous, gather, print who took part; no planning")
    return ap.parse_args()


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(a) -> int:
    """`bench.py --gpus N` without a launcher: become the launcher (one process per GPU, torch.distributed.run)."""
    if not a.rehearse:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < a.gpus and not os.environ.get("BENCH_SHARE_GPU"):
            print(f"bench.py --gpus {a.gpus}: only {have} GPU(s) visible; one process per GPU needs {a.gpus}", file=sys.stderr)
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


_STDOUT_FD = None


def emit(obj) -> None:
    """The one JSON line, on the process's real stdout."""
    line = (json.dumps(obj) + "\n").encode()
    sys.stdout.flush()
    os.write(_STDOUT_FD if _STDOUT_FD is not None else 1, line)


def host_cpu():
    model, phys = "unknown", set()
    try:
        pid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                phys.add((pid, line.split(":", 1)[1].strip()))
    except OSError:
        pass
    return {"model": model, "logical_cpus": os.cpu_count(), "physical_cores": len(phys) or None}


def pick_device(local, ndev, world, share):
    """(device ordinal, ranks share a GPU, identities must be checked) for a rank.  Three launch shapes: every rank sees all GPUs
    (LOCAL_RANK picks); every rank sees exactly one GPU of its own (a launcher that sets ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES
    per rank: device 0, and the ranks compare identities once the process group is up); a 1-GPU box rehearsing (BENCH_SHARE_GPU)."""
    if ndev <= 0:
        return None, False, False
    if ndev >= world or (local < ndev and ndev > 1):
        return (local, False, False) if local < ndev else (None, False, False)
    if share:
        return local % ndev, True, False
    if ndev == 1 and world > 1:
        return 0, False, True                            # EVERY rank of such a launch checks (rank 0 included: the check is a collective)
    return None, False, False


def gpu_identity(torch, dev):
    pr = torch.cuda.get_device_properties(dev)
    for key in ("uuid", "pci_bus_id"):
        v = getattr(pr, key, None)
        if v is not None:
            return f"{key}:{v}:{getattr(pr, 'pci_domain_id', '')}:{getattr(pr, 'pci_device_id', '')}"
    return f"name:{pr.name}:{os.environ.get('ROCR_VISIBLE_DEVICES', '')}:{os.environ.get('HIP_VISIBLE_DEVICES', '')}"


def distinct_gpus(ids):
    return len(set(ids)) == len(ids)


def main():
    a = parse()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(a))
    # Exactly ONE line on stdout: libraries that write to file descriptor 1 themselves (RCCL prints a version banner there when a
    # process group comes up) go to stderr from here on; the JSON line is written to the saved descriptor at the end.
    global _STDOUT_FD
    sys.stdout.flush()
    _STDOUT_FD = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus} was launched with WORLD_SIZE={world}: they must agree")
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")     # "nccl" is RCCL on ROCm; "gloo" only for rehearsals
    if a.rehearse:
        return rehearse(rank, world, local, backend)

    import numpy as np
    import torch
    from benchnav_amd import NativeMPPI, _capi, synth
    from benchnav_amd.sharding import gather_times

    host_threads = torch.get_num_threads()
    # The GPU legs need no host arithmetic: keep torch's intra-op pool out of the process until the CPU baseline leg
    # (spinning pool threads next to the HIP runtime's submission thread were seen to stretch a timed loop 2x).
    torch.set_num_threads(1)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the MPPI planner has no CPU fallback")
    ndev = torch.cuda.device_count()
    dev, shared_gpu, verify_distinct = pick_device(local, ndev, world, bool(os.environ.get("BENCH_SHARE_GPU")))
    if dev is None:
        raise SystemExit(f"rank {rank}: local rank {local} has no GPU ({ndev} visible); one process per GPU needs {world}")
    torch.cuda.set_device(dev)
    dist = None
    # BENCH_FORCE_DIST=1: a single rank goes through the collective path as well (RCCL group of one: init, barrier, device
    # all-gather, object gather) -- what lets a 1-GPU box execute the code an 8-GPU launch runs
    if world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist_
        dist = dist_
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev), rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    coll_dev = torch.device("cuda", dev) if (dist is not None and backend == "nccl") else None
    if verify_distinct and dist is not None:
        # every rank sees exactly ONE device (a launcher that sets ROCR_/HIP_VISIBLE_DEVICES per rank): they must be different GPUs
        ids = [None] * world
        dist.all_gather_object(ids, gpu_identity(torch, dev))
        if not distinct_gpus(ids):
            print(f"rank {rank}: the ranks do not see {world} different GPUs ({ids}); set BENCH_SHARE_GPU=1 to rehearse on a shared one", file=sys.stderr)
            sys.stderr.flush()
            os._exit(2)                                  # every rank decides the same; no collective is left to hang in

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    stream = torch.cuda.current_stream()
    if a.workload != "c2":
        out = run_workload(a, a.workload, a.steps, rank, world, dev, dist, coll_dev, sync, stream, shared_gpu, backend, ndev)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            emit(out)
        return

    def make_planner(inst, B=1, **kw):
        kw.setdefault("overlap", not a.no_overlap)
        pl = NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=RES, num_instances=B, device_id=dev,
                        stream=stream.cuda_stream, **kw)
        pl.set_map(inst.risk.numpy())
        pl.set_goal(inst.goal.numpy())
        return pl

    def timed_solves(pl, state_dev, eps_ring, kind, steps, sync_, events=True):
        """Enqueue `steps` dependent solves (software-pipelined: one launch each), then the tail of the last one.
        Returns (wall seconds between the two syncs, milliseconds between HIP events placed on the launch stream before
        the first and after the last launch, or None without `events`).  Every solve's U*, X* and weights are written."""
        if events:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sync_()
        t0 = time.perf_counter()
        if events:
            e0.record(stream)
        if eps_ring is None:
            pl.solve_n_async_device(steps, state_dev.data_ptr())
        else:                                            # eps_ring: one contiguous tensor (ring, ...), cycled per solve
            pl.solve_n_async_device(steps, state_dev.data_ptr(), eps_ring.data_ptr(), kind, eps_ring.shape[0], eps_ring[0].numel())
        if events:
            e1.record(stream)
        pl.flush()
        torch.cuda.synchronize()                         # this rank's K steps are complete ...
        wall = time.perf_counter() - t0
        if sync_ is sync and dist is not None:
            dist.barrier()                               # ... and the ranks meet again behind the clock (see the module docstring)
        pl.sync()            # outside the timed region: the library's own synchronisation point (checks the overlapped launches' error word)
        return wall, (e0.elapsed_time(e1) if events else None)

    def settle(make, state_dev, eps_ring, kind, warmup, tries=3):
        """Untimed: build the planner, run the warm-up steps, and make sure the queue is not in the rare slow-dispatch
        state seen on some boxes (milliseconds between back-to-back launches): if a 100-step probe is >5x slower than the
        best probe seen, rebuild the handle and try again."""
        best = None
        for _ in range(tries):
            pl = make()
            timed_solves(pl, state_dev, eps_ring, kind, max(warmup, 1), torch.cuda.synchronize)
            probes = [timed_solves(pl, state_dev, eps_ring, kind, 100, torch.cuda.synchronize)[0] / 100 for _ in range(3)]
            p = min(probes)
            if os.environ.get("BENCH_DEBUG"):
                print(f"[settle] probes us/step: {[round(x * 1e6, 2) for x in probes]}", file=sys.stderr)
            best = p if best is None else min(best, p)
            if p <= 5 * best and p < 2e-3:
                return pl
            pl.close()
        return make()

    def repeated(pl, state_dev, eps_ring, kind, steps, sync_, repeats):
        """The contract's timed region, `repeats` times: nothing inside it but the K solves and the last tail.  The kernel's
        launch-to-launch time comes from the same handle right afterwards: the same K launches between two HIP events."""
        walls = [timed_solves(pl, state_dev, eps_ring, kind, steps, sync_, events=False)[0] for _ in range(repeats)]
        evs = [timed_solves(pl, state_dev, eps_ring, kind, steps, torch.cuda.synchronize)[1] for _ in range(min(repeats, 20))]
        return walls, evs

    inst = synth.make_instance(G, seed=rank, resolution=RES)       # independent map seed per rank
    state_dev = inst.start.cuda()
    if a.noise == "injected":
        gen = torch.Generator(device="cuda").manual_seed(1234 + rank)
        eps_ring = torch.randn(8, T, 2, K, device="cuda", generator=gen)
        kind = _capi.BN_NOISE_DEVICE_T2K
    else:
        eps_ring, kind = None, _capi.BN_NOISE_PHILOX
    injected = a.noise == "injected"

    # host-side instance generation for the batched leg happens before any timing (no idle gap later)
    extras = rank == 0 and world == 1 and not a.no_extras
    batched_insts = None
    if extras and not a.no_batched:
        batched_insts = [synth.make_instance(G, seed=s, resolution=RES, jitter=True) for s in range(a.batched_instances)]

    # ---- headline: dependent solves of one instance per GPU, K steps x `repeats` ------------------------------
    repeats = max(1, min(200, -(-MIN_TIMED_STEPS // max(a.steps, 1))))
    pl = settle(lambda: make_planner(inst), state_dev, eps_ring, kind, a.warmup)
    walls, evs = repeated(pl, state_dev, eps_ring, kind, a.steps, sync, repeats)
    alg_bytes = pl.algorithmic_bytes(injected_noise=injected)
    sustained = None
    if rank == 0 and world == 1 and a.steps < 1000:      # the steady-state figure next to a short contract run
        w_s, e_s = timed_solves(pl, state_dev, eps_ring, kind, 3000, torch.cuda.synchronize)
        sustained = {"steps": 3000, "value": 3000 / w_s, "ms_per_step": w_s / 3000 * 1e3, "kernel_ms": e_s / 3000}
    pl.close()
    per_rank = gather_times(walls, coll_dev)                       # (world, repeats): the only data exchange, 8 B x repeats per rank
    job = per_rank.max(dim=0).values                               # the slowest rank bounds every repeat
    med = float(job.median())
    r_med = int((job - med).abs().argmin())
    value = world * a.steps / med
    names = [None] * world
    me = f"rank {rank}: cuda:{dev} {torch.cuda.get_device_name(dev)}"
    if dist is not None:
        dist.all_gather_object(names, me)
    else:
        names = [me]

    out = None
    if rank == 0:
        kernel_ms = statistics.median(evs) / a.steps               # same handle, the same K launches, right after the timed repeats
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")     # PMC-derived HBM bytes per launch, see profiles/README.md
        tj = {}
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
            except Exception:
                tj = {}
        traffic = tj.get(f"rollout_{a.noise}_B1")
        traffic_src = tj.get("collected") if traffic is not None else None
        # measured device-to-device copy rate of this box (1 GiB read + 1 GiB written per pass): the practical HBM ceiling
        src = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
        dst = torch.empty_like(src)
        dst.copy_(src)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            dst.copy_(src)
        torch.cuda.synchronize()
        copy_gbs = 10 * 2 * src.numel() * 4 / (time.perf_counter() - t0) / 1e9
        del src, dst
        out = {
            "metric": "MPPI solve-steps/sec (K=1024,T=50,256x256 map)", "value": value, "unit": "solves/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": med / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "repeats": repeats, "ms_per_step_min": float(job.min()) / a.steps * 1e3, "ms_per_step_max": float(job.max()) / a.steps * 1e3,
            "world_size": world, "devices": names, "collective_backend": (dist.get_backend() if dist is not None else None),
            "per_rank_solves": [a.steps] * world, "per_rank_seconds": [float(x) for x in per_rank[:, r_med]],
            "host_cpu": host_cpu(),
            "config": {"workload": "BASELINE configs[1]: single 256x256 map, K=1024, T=50, one instance per GPU, "
                                   "dependent warm-started solves, fixed state",
                       "grid": G, "num_samples": K, "horizon": T, "resolution": RES, "instances_per_gpu": 1,
                       "arithmetic": "spec (default: carried heading vector, fused transit; DESIGN.md 5.1) -- `value_reference_order` is the "
                                     "same workload in the reference's operation order (BN_FLAG_REFERENCE_ORDER)",
                       "noise": "philox in-kernel (sampling inside the timed region)" if a.noise == "philox"
                                else "injected eps (T,2,K) resident in HBM",
                       "parallelism": f"instance sharding x{world}, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "peak_measured_copy": copy_gbs, "frac_of_measured_copy": achieved / copy_gbs,
                         "kernel": "bn::rollout_lat_kernel (barrier-free variant of the 5-wave role kernel, one workgroup per CU; "
                                   "every launch also carries the previous solve's merge + tail workgroup)",
                         "kernel_ms": kernel_ms,
                         "kernel_ms_source": "HIP event pair on the launch stream around the same K launches on the same handle, right after "
                                             "the timed repeats (median; launch-to-launch mean).  With overlapped launches (default) two kernels "
                                             "are in flight and each one's own duration, as rocprofv3 lists it, includes its wait for the "
                                             "predecessor's partials (about twice this figure); `no_overlap` below is the one-stream "
                                             "configuration whose per-kernel duration rocprofv3 reports directly",
                         "overlapped_launches": not a.no_overlap,
                         "algorithmic_bytes_per_launch": alg_bytes, "launches_timed": a.steps,
                         "note": "single-instance solve = 16 workgroups x a 50-step serial chain (~5.2 us of dependent instructions): "
                                 "latency-bound; at 0.9 MB per solve one launch boundary (~1.4 us) alone caps a launch-per-solve design "
                                 "at ~7 % of HBM peak and the chain at ~2 %; the batched object is the HBM-relevant regime (DESIGN.md 6)"},
        }
        if not a.no_overlap:
            pn = make_planner(inst, overlap=False)
            timed_solves(pn, state_dev, eps_ring, kind, max(a.warmup, 50), torch.cuda.synchronize)
            w_n, e_n = timed_solves(pn, state_dev, eps_ring, kind, max(a.steps, 1000), torch.cuda.synchronize)
            pn.close()
            n_n = max(a.steps, 1000)
            out["no_overlap"] = {"value": n_n / w_n, "unit": "solves/s", "ms_per_step": w_n / n_n * 1e3, "kernel_ms": e_n / n_n, "steps": n_n,
                                 "roofline_frac": alg_bytes / (e_n / n_n * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "note": "BN_FLAG_NO_OVERLAP: every launch on the handle's stream, one kernel in flight"}
        if shared_gpu or backend != "nccl":
            out["rehearsal"] = f"ranks shared {ndev} GPU(s), backend {backend}: not a scaling measurement"
        if sustained:
            out["sustained"] = sustained
            out["sustained"]["note"] = ("the contract's K steps are one short timed region: launch latency of the first solve, the last solve's tail "
                                        "kernel and the synchronisation are paid once per region; this is the same handle over 3000 steps")
            out["fixed_overhead_us_per_timed_region"] = (med / a.steps - sustained["ms_per_step"] * 1e-3) * a.steps * 1e6
            # what the runtime alone needs for the same bracket around ONE trivial kernel (torch's fill of one element): launch latency
            # + completion -> host; the part of the fixed cost above that no kernel change can remove (tools/spin_sync.py: the
            # hipDeviceSchedule* flags do not move it)
            one = torch.zeros(1, device="cuda")
            fl = []
            for _ in range(200):
                torch.cuda.synchronize(); t0 = time.perf_counter(); one.fill_(1.0); torch.cuda.synchronize(); fl.append(time.perf_counter() - t0)
            out["runtime_floor_us_one_kernel_region"] = sorted(fl)[len(fl) // 2] * 1e6

    if extras:
        def leg(pl_, st_, ring_, n_):
            if os.environ.get("BENCH_DEBUG"):
                print(f"[leg] B={pl_.B} K={pl_.K} T={pl_.T} n={n_}", file=sys.stderr, flush=True)
            timed_solves(pl_, st_, ring_, kind, 30, torch.cuda.synchronize)
            best = None
            for _ in range(3):
                w, e = timed_solves(pl_, st_, ring_, kind, n_, torch.cuda.synchronize)
                if best is None or w < best[0]:
                    best = (w, e)
            return best[0] / n_, best[1] / n_              # seconds per launch (wall), ms per launch (events)

        def roof(bytes_, ms, traffic_key, window_bytes=None):
            r_ = {"bound": "hbm", "achieved": bytes_ / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": bytes_ / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": tj.get(traffic_key), "kernel_ms": ms,
                  "algorithmic_bytes_per_launch": bytes_}
            if window_bytes is not None:
                # SURVEY 8d counts the whole map (4 G^2) per instance; the kernels stage only the window a rollout can reach within the
                # horizon.  With the trajectory dump that is a few per cent of the bytes, in lean mode most of them: the fraction against
                # the window figure is the one that says how much of the roofline the launch's own traffic reaches
                r_["algorithmic_bytes_window_per_launch"] = window_bytes
                r_["frac_window"] = window_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            return r_

        # ---- lean mode, single instance: no trajectory dump (SURVEY 8d lean formula) ---------------------------
        pll = make_planner(inst, lean=True)
        s_l, ms_l = leg(pll, state_dev, eps_ring, max(500, a.steps))
        by_l = pll.algorithmic_bytes(injected_noise=injected)
        by_lw = pll.algorithmic_bytes(injected_noise=injected, window=True)
        pll.close()
        out["lean"] = {"value": 1.0 / s_l, "unit": "solves/s", "ms_per_step": s_l * 1e3,
                       "note": "BN_FLAG_LEAN: _state_seq_batch not materialised; get_top_samples re-rolls the requested rows bit-identically",
                       "roofline": roof(by_l, ms_l, f"rollout_{a.noise}_B1_lean", by_lw)}
        # ---- BN_FLAG_REFERENCE_ORDER: the transit in the reference's own operation order (same kernels, one launch per solve: `launches_per_solve`) ----
        plo = make_planner(inst, reference_order=True)
        s_o, ms_o = leg(plo, state_dev, eps_ring, max(300, a.steps))
        assert plo.arithmetic() == "reference_order"
        lps_o = plo.launches_per_solve()
        plo.close()
        out["reference_order"] = {"value": 1.0 / s_o, "unit": "solves/s", "us_per_solve": s_o * 1e6, "launches_per_solve": lps_o,
                                  "slowdown_vs_default": s_o / (med / a.steps) if not sustained else s_o / (sustained["ms_per_step"] * 1e-3),
                                  "note": "same workload with BN_FLAG_REFERENCE_ORDER: sincos of every step's heading and x + ((trav v) cos) dt as "
                                          "robot_model.py:86-88 writes it, on the same kernels (rollout_role_ref_*.hip: the chain wave integrates the "
                                          "heading itself); see parity_census for what it buys.  Batched launches and the ticket paths pay 1-8 % for it "
                                          "(tools/ref_rate.py), this single-instance latency path ~38 %"}
        # Which number is the parity number.  `value` above is measured in the default arithmetic (config.arithmetic), which keeps
        # SURVEY 8a's trajectory tolerance (i) for all but ~3e-5 of rollouts (parity_census: every exception is a counted cell flip);
        # the arithmetic that meets (i) outright is this one, on the same workload, sustained over the same number of dependent solves
        out["value_reference_order"] = 1.0 / s_o
        # ---- batched: 64 instances per launch on this GPU (HBM-relevant regime) -------------------
        if not a.no_batched:
            B = a.batched_instances
            insts = batched_insts
            states = torch.stack([it.start for it in insts]).cuda()
            ring = torch.randn(2, B, T, 2, K, device="cuda") if injected else None
            nb = max(400, a.steps // 5)
            res_b = {}
            for lean, overlap in ((False, not a.no_overlap), (True, not a.no_overlap)) + (() if a.no_overlap else ((False, False),)):
                plb = NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=RES, num_instances=B, device_id=dev,
                                 stream=stream.cuda_stream, lean=lean, overlap=overlap)
                for b, it in enumerate(insts):
                    plb.set_map(it.risk.numpy(), b)
                    plb.set_goal(it.goal.numpy(), b)
                s_b, ms_b = leg(plb, states, ring, nb)
                bytes_b = plb.algorithmic_bytes(injected_noise=injected) * B
                bytes_bw = plb.algorithmic_bytes(injected_noise=injected, window=True) * B
                plb.close()
                r_ = {"value": B / s_b, "unit": "solves/s", "ms_per_launch": s_b * 1e3, "overlapped_launches": overlap,
                      "roofline": roof(bytes_b, ms_b, f"rollout_{a.noise}_B{B}" + ("_lean" if lean else ""), bytes_bw)}
                if overlap != (not a.no_overlap):
                    r_["note"] = ("every launch on one stream: the configuration whose per-kernel duration rocprofv3 reports "
                                  "(profiles/*_kernel_stats_no_overlap.csv); with overlapped launches a kernel's duration includes "
                                  "the time its workgroups wait for their instance's previous solve, so kernel_ms there is the "
                                  "launch-to-launch time between HIP events")
                    res_b["no_overlap"] = r_
                else:
                    res_b[lean] = r_
            # the same 64-instance launch in the reference's operation order
            plq = NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=RES, num_instances=B, device_id=dev,
                             stream=stream.cuda_stream, overlap=not a.no_overlap, reference_order=True)
            for b, it in enumerate(insts):
                plq.set_map(it.risk.numpy(), b)
                plq.set_goal(it.goal.numpy(), b)
            s_q, ms_q = leg(plq, states, ring, nb)
            plq.close()
            out["reference_order"]["batched"] = {"instances_per_launch": B, "value": B / s_q, "unit": "solves/s", "ms_per_launch": s_q * 1e3,
                                                 "slowdown_vs_default": s_q / (B / res_b[False]["value"])}
            # a launch large enough for the one-wave throughput kernel (auto-selected above ~1900 workgroups)
            BL = 256
            large = {}
            for lean in (False, True):
                plw = make_planner(inst, B=BL, shared_map=True, lean=lean)
                stl = torch.stack([inst.start] * BL).cuda()
                s_w, ms_w = leg(plw, stl, None, max(200, a.steps // 10))
                bytes_w = plw.algorithmic_bytes(injected_noise=False) * BL
                bytes_ww = plw.algorithmic_bytes(injected_noise=False, window=True) * BL
                plw.close()
                large[lean] = {"instances_per_launch": BL, "value": BL / s_w, "unit": "solves/s", "ms_per_launch": s_w * 1e3,
                               "kernel": "bn::rollout_wave_park_kernel (one wave per 64 rollouts; controls parked in registers)",
                               "roofline": roof(bytes_w, ms_w, f"rollout_wave_{a.noise}_B{BL}" + ("_lean" if lean else ""), bytes_ww)}
            large[False]["lean"] = large[True]
            out["batched"] = dict(res_b[False], instances_per_launch=B, lean=res_b[True], large_batch=large[False])
            # what ONE of 8 GPUs runs for BASELINE configs[3] (64 instances sharded 8 per GPU, `--workload c4 --gpus 8`): 8 instances
            # per launch.  136 workgroups: every one has a CU to itself (latency kernel), so the leg is the single-instance chain
            # latency amortised over 8 instances -- what an 8-GPU run's per-GPU rate can be at best
            B8 = 8
            pl8 = NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=RES, num_instances=B8, device_id=dev,
                             stream=stream.cuda_stream, overlap=not a.no_overlap)
            for b, it in enumerate(insts[:B8]):
                pl8.set_map(it.risk.numpy(), b)
                pl8.set_goal(it.goal.numpy(), b)
            s_8, ms_8 = leg(pl8, states[:B8].contiguous(), (ring[:, :B8].contiguous() if ring is not None else None), max(1000, a.steps))
            bytes_8 = pl8.algorithmic_bytes(injected_noise=injected) * B8
            bytes_8w = pl8.algorithmic_bytes(injected_noise=injected, window=True) * B8
            pl8.close()
            out["per_gpu_share_c4"] = {"instances_per_launch": B8, "value": B8 / s_8, "unit": "solves/s", "ms_per_launch": s_8 * 1e3,
                                       "projected_8_gpu_value": 8 * B8 / s_8,
                                       "note": "BASELINE configs[3] sharded over 8 GPUs = 8 instances per launch and GPU; measured on ONE GPU. "
                                               "projected_8_gpu_value = 8 x this rate (instance sharding has no data-path collective): a projection, "
                                               "not a measurement -- no multi-GPU node was available to the builder",
                                       "roofline": roof(bytes_8, ms_8, f"rollout_{a.noise}_B{B8}", bytes_8w)}
            if "no_overlap" in res_b:
                out["batched"]["no_overlap"] = res_b["no_overlap"]
        # ---- the reference's own boundary: MPPI.forward(state) once per control step, host consuming action_seq[0] -------------
        out["dropin_forward"] = dropin_forward(inst, dev, max(a.steps, 1000))
        out["value_dropin_forward"] = out["dropin_forward"]["value"]
        # ---- closed loop on the device: solve -> PlanetaryEnv.step -> solve ..., one launch per control step ----
        plc = make_planner(inst)
        plc.env_attach(inst.risk.numpy(), np.full((G, G), 0.05, np.float32))      # latent slip ~ N(risk, 0.05)
        n_cl = max(a.steps, 1000)
        plc.episode(200, inst.start.numpy())
        elc = None
        for _ in range(3):          # best of three (like the other legs)
            t0 = time.perf_counter()
            states, rewards, done = plc.episode(n_cl, inst.start.numpy())
            dt_ = time.perf_counter() - t0
            elc = dt_ if elc is None else min(elc, dt_)
        plc.close()
        out["closed_loop"] = {"value": n_cl / elc, "unit": "control steps/s", "us_per_step": elc / n_cl * 1e6,
                              "instances": 1, "steps": n_cl,
                              "distance_to_goal_m": [float(np.linalg.norm(states[0, 0, :2] - inst.goal.numpy())),
                                                     float(np.linalg.norm(states[-1, 0, :2] - inst.goal.numpy()))],
                              "note": "bn_mppi_episode_async: state advanced on the device by the observation-mode "
                                      "transit with sampled slip (planetary_env.py:189-219); includes the final log copy"}
        # ---- the operating point the reference itself states (test/test_mppi.py:121-169, tutorial 3.3): K=5000, T=50, 64x64 map at
        # 0.5 m, CVaR-0.9 risk map, start (8,8) heading at the goal (24,24).  K = 5000 is ragged and above the ticket-merge switch ----
        from benchnav_amd.risk import infer_risk_map
        Kr, Gr = 5000, 64
        risk_r = infer_risk_map(synth.smooth_risk_map(Gr, 9) * 0.7, synth.slip_std_map(Gr, 9), "cvar", 0.9, seed=0).cpu().numpy()
        plr = NativeMPPI(horizon=T, num_samples=Kr, grid_size=Gr, resolution=RES, device_id=dev, stream=stream.cuda_stream)
        plr.set_map(risk_r); plr.set_goal(np.array([24.0, 24.0], np.float32))
        st_r = torch.tensor([8.0, 8.0, 0.7853981633974483], device="cuda")
        s_r, ms_r = leg(plr, st_r, None, max(200, a.steps // 5))
        bytes_r = plr.algorithmic_bytes(injected_noise=False)
        plr.close()
        out["reference_operating_point"] = {
            "workload": f"the reference's own test / tutorial configuration (test/test_mppi.py:121-169): mppi_solve K={Kr} T={T} map={Gr}x{Gr} "
                        f"at {RES} m, CVaR-0.9 risk map (risk-map kernel), dependent warm-started solves, fixed state",
            "value": 1.0 / s_r, "unit": "solves/s", "us_per_solve": s_r * 1e6,
            "roofline": dict(roof(bytes_r, ms_r, f"rollout_K{Kr}_T{T}_G{Gr}"),
                             kernel="bn::rollout_kernel (role kernel, ticket merge by the last workgroup; previous tail as aux workgroup)"),
            "parity": "tests/golden/ref5000.npz: three warm-started solves of the imported reference at this configuration, "
                      "teacher-forced and free-running (tests/test_gpu_parity.py)"}
        # ---- BASELINE config 3: K=8192, T=50, slip sampled per lookup from Normal(mean, std) (Philox in-kernel) ----
        K3 = 8192
        pl3 = NativeMPPI(horizon=T, num_samples=K3, grid_size=G, resolution=RES, device_id=dev, sampled_slip=True, stream=stream.cuda_stream)
        pl3.set_map(inst.risk.numpy()); pl3.set_slip_std(synth.slip_std_map(G, seed=0).numpy()); pl3.set_goal(inst.goal.numpy())
        s3, ms3 = leg(pl3, state_dev, None, max(100, a.steps // 10))
        pl3.close()
        # algorithmic bytes (SURVEY.md 8d with in-kernel draws): mean + std maps, X, cost + weights, mean/state/U*/X*
        bytes3 = 8 * G * G + 12 * K3 * (T + 1) + 8 * K3 + 28 * T + 24
        out["sampled_slip"] = {"workload": f"BASELINE configs[2]: mppi_solve K={K3} T={T} map={G}x{G}, slip ~ Normal(mean, std)[cell] drawn per lookup",
                               "value": 1.0 / s3, "unit": "solves/s", "us_per_solve": s3 * 1e6, "draws_per_solve": K3 * (2 * T + 1) + T,
                               "roofline": dict(roof(bytes3, ms3, f"rollout_sampled_K{K3}"),
                                                kernel="bn::rollout_sampled_kernel (one launch per solve: rollouts, ticket merge, previous tail)")}
        # ---- BASELINE configs[4] on one GPU: 512x512 map, K=16384, T=100 (LDS-tiled 43x43 window) ----
        K5, T5, G5 = 16384, 100, 512
        inst5 = synth.make_instance(G5, seed=0, resolution=RES)
        pl5 = NativeMPPI(horizon=T5, num_samples=K5, grid_size=G5, resolution=RES, device_id=dev, stream=stream.cuda_stream)
        pl5.set_map(inst5.risk.numpy()); pl5.set_goal(inst5.goal.numpy())
        st5 = inst5.start.cuda()
        s5, ms5 = leg(pl5, st5, None, max(50, a.steps // 20))
        bytes5 = pl5.algorithmic_bytes(injected_noise=False)
        pl5.close()
        out["config5_one_gpu"] = {"workload": f"BASELINE configs[4] on one GPU: mppi_solve K={K5} T={T5} map={G5}x{G5}",
                                  "value": 1.0 / s5, "unit": "solves/s", "us_per_solve": s5 * 1e6,
                                  "roofline": roof(bytes5, ms5, f"rollout_K{K5}_T{T5}_G{G5}")}
    if world > 1 and not a.no_extras:
        # A multi-GPU run of the DEFAULT command: the two BASELINE configurations that name 8 GPUs ride along as objects of the
        # same line (configs[3]: 64 instances sharded 64/N per GPU; configs[4]: one K=16384 solve sharded by rollouts), each with
        # its own roofline and per-rank times -- what `--workload c4|c5` prints as a line of its own.  Every rank takes part.
        # ... under a watchdog: these riders are the only part of a multi-GPU run that has never executed on more than one GPU.  Should
        # a collective inside them hang, every rank gives up after 120 s and rank 0 prints the line it has (the headline above) --
        # a rider must not be able to take the measurement down with it.
        import threading

        def give_up():
            if rank == 0:
                out.setdefault("sharded_c4", {"error": "timed out"}); out.setdefault("sharded_c5", {"error": "timed out"})
                out["parity_census"] = parity_census()
                emit(out)
            os._exit(0)
        dog = threading.Timer(120.0, give_up)
        dog.daemon = True
        dog.start()
        for which in ("c4", "c5"):
            try:
                o = run_workload(a, which, min(a.steps, 50), rank, world, dev, dist, coll_dev, sync, stream, shared_gpu, backend, ndev)
                if rank == 0:
                    out["sharded_" + which] = {k: o[k] for k in ("value", "unit", "ms_per_step", "steps", "repeats", "scaling", "per_rank_seconds", "config", "roofline")}
            except Exception as e:                           # the headline above stands whatever happens to a rider
                print(f"rank {rank}: sharded_{which} failed: {e!r}", file=sys.stderr)
                if rank == 0:
                    out["sharded_" + which] = {"error": repr(e)[:500]}
        dog.cancel()
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(inst, a.cpu_seconds, host_threads)
    if rank == 0:
        out["parity_census"] = parity_census()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit(out)


def dropin_forward(inst, dev, n_steps):
    """The path a BenchNav user gets after changing the import: `benchnav_amd.MPPI(...)` built from reference-shaped `dynamics` /
    `objectives` objects and driven as the reference's loop drives it (test/test_mppi.py:174-181) -- one `forward(state)` per control
    step, the state living on the HOST and changing every step, the host consuming `action_seq[0]` before it can take the next step.
    Per step: forward() (ONE launch on the latency kernel: rollouts and the solve's own tail, bn_mppi_forward_state_async; with
    host_loop=True that launch was enqueued one step ahead and waits on the device for the state this call posts to pinned memory),
    first_action() (polls the tail's pinned-memory mailbox: no stream synchronisation, no copy), and a
    host-side environment step (the unicycle transit of planetary_env.py:203-205 on three Python floats with a nearest-cell lookup in the
    host copy of the risk map).  Also timed: the unmodified reference loop's read-back, `action_seq[0].cpu()`."""
    import math
    import types
    import numpy as np
    import torch
    from benchnav_amd import MPPI
    c = G * RES / 2
    gm = types.SimpleNamespace(grid_size=G, resolution=RES, x_limits=(c - G / 2 * RES, c + G / 2 * RES), y_limits=(c - G / 2 * RES, c + G / 2 * RES))
    dyn = types.SimpleNamespace(_grid_map=gm, _traversability_model=types.SimpleNamespace(_risks=inst.risk), _model_config=types.SimpleNamespace(mode="inference"),
                                min_action=torch.tensor([0.0, -1.0]), max_action=torch.tensor([1.0, 1.0]))
    obj = types.SimpleNamespace(_goal_pos=inst.goal, _stuck_threshold=0.3, stage_cost=None, terminal_cost=None)
    def make(host_loop):
        return MPPI(horizon=T, num_samples=K, dim_state=3, dim_control=2, dynamics=dyn, objectives=obj, sigmas=torch.tensor([0.5, 0.5]), lambda_=0.5,
                    device=torch.device("cuda", dev), seed=42, noise="philox", store_controls=False, host_loop=host_loop)
    solver = solver_paced = make(True)
    risk = inst.risk.numpy()
    state = inst.start.clone()                           # a CPU tensor, rewritten in place every step (forward() takes it by value)
    sv = state.numpy()
    s0 = sv.copy()
    hi = G * RES

    def env_step(a0, a1):                                # PlanetaryEnv.step's transit (robot_model.py:75-95) on the host, dt = 0.1
        x, y, th = float(sv[0]), float(sv[1]), float(sv[2])
        trav = 1.0 - min(max(float(risk[min(max(int(y / RES), 0), G - 1), min(max(int(x / RES), 0), G - 1)]), 0.0), 1.0)
        sv[0] = min(max(x + trav * a0 * math.cos(th) * 0.1, 0.0), hi)
        sv[1] = min(max(y + trav * a0 * math.sin(th) * 0.1, 0.0), hi)
        sv[2] = (th + trav * a1 * 0.1 + math.pi) % (2 * math.pi) - math.pi

    def loop(n, readback, solver=None):
        solver = solver or solver_paced
        t_f = t_a = t_e = 0.0
        sv[:] = s0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            if i % 75 == 0:
                sv[:] = s0                               # episodes of 75 control steps (the goal is ~85 steps away): the rollouts keep their reach
            ta = time.perf_counter()
            U, X = solver.forward(state)
            tb = time.perf_counter()
            act = U[0].cpu() if readback else solver.first_action()
            a0, a1 = act.tolist()                        # (one call: two tensor indexings cost more than the environment step)
            tc = time.perf_counter()
            env_step(a0, a1)
            td = time.perf_counter()
            t_f += tb - ta; t_a += tc - tb; t_e += td - tc
        wall = time.perf_counter() - t0
        solver.release()                                 # ends the loop: a launch waiting for the next state is cancelled (host_loop)
        torch.cuda.synchronize()
        return wall / n, t_f / n, t_a / n, t_e / n

    loop(200, False)
    best = min((loop(n_steps, False) for _ in range(3)), key=lambda r: r[0])
    rb = min((loop(max(n_steps // 2, 300), True) for _ in range(2)), key=lambda r: r[0])
    plain = make(False)                                  # the same loop without the opt-in: one ordinary launch per forward()
    loop(200, False, plain)
    one = min((loop(n_steps, False, plain) for _ in range(3)), key=lambda r: r[0])
    del plain
    acts = make("actions")                               # the opt-in's leaner form: forward() does not order torch's stream behind the solve
    loop(200, False, acts)
    ao = min((loop(n_steps, False, acts) for _ in range(3)), key=lambda r: r[0])
    U, X = acts.forward(state)
    fa = acts.first_action().clone()
    acts.order_outputs()
    same_ao = bool(torch.equal(fa, U[0].cpu()))
    acts.release()
    del acts
    lpf = int(solver._lib.bn_mppi_launches_per_forward(solver._handle))
    # cross-check of what the loop consumed: the mailbox value IS action_seq[0]
    U, X = solver.forward(state)
    same = bool(torch.equal(solver.first_action(), U[0].cpu()))
    return {"value": 1.0 / best[0], "unit": "control steps/s", "us_per_step": best[0] * 1e6, "steps": n_steps,
            "split_us": {"forward_call_host": best[1] * 1e6, "first_action_wait": best[2] * 1e6, "host_env_step": best[3] * 1e6,
                         "loop_overhead": (best[0] - best[1] - best[2] - best[3]) * 1e6},
            "launches_per_forward": lpf, "first_action_equals_action_seq0": same, "host_loop": bool(solver._host_loop),
            "without_host_loop": {"value": 1.0 / one[0], "us_per_step": one[0] * 1e6,
                                  "split_us": {"forward_call_host": one[1] * 1e6, "first_action_wait": one[2] * 1e6, "host_env_step": one[3] * 1e6},
                                  "note": "MPPI(host_loop=False), the default: one launch per forward() (rollouts + the solve's own tail), enqueued when forward() is called"},
            "host_loop_actions": {"value": 1.0 / ao[0], "us_per_step": ao[0] * 1e6,
                                  "split_us": {"forward_call_host": ao[1] * 1e6, "first_action_wait": ao[2] * 1e6, "host_env_step": ao[3] * 1e6},
                                  "first_action_equals_action_seq0": same_ao,
                                  "note": "MPPI(host_loop='actions'), BN_FLAG_UNORDERED_OUTPUTS: forward() leaves torch's stream unordered behind the solve "
                                          "(the stream-wait is 3-5 us of host time per step); the planner's own attributes and order_outputs() make up for it on demand"},
            "with_cpu_readback": {"value": 1.0 / rb[0], "us_per_step": rb[0] * 1e6,
                                  "note": "the unmodified reference loop: action_seq[0].cpu() instead of first_action() -- a stream synchronisation and a copy"},
            "config": {"class": "benchnav_amd.MPPI (drop-in for src/planners/local_planners/mppi.py:MPPI)", "noise": "philox", "copy_outputs": True, "host_loop": True,
                       "state": "host (CPU tensor), new every control step", "workload": "BASELINE configs[1]: K=1024, T=50, 256x256"},
            "note": "one solve per step with the host in the loop: launch latency, the whole solve's latency and the mailbox's trip to the host are "
                    "paid every step, nothing overlaps -- the reference boundary's figure; `value` above is the device-side chain of dependent solves"}


def run_workload(a, which, steps, rank, world, dev, dist, coll_dev, sync, stream, shared_gpu, backend, ndev):
    """--workload c4 / c5: the two BASELINE configurations that name 8 GPUs, same contract line (metric, value = whole-job rate,
    barrier + synchronize around exactly K steps, max over ranks), one `roofline` per workload."""
    import numpy as np
    import torch
    from benchnav_amd import NativeMPPI, synth
    from benchnav_amd.sharding import ShardedMPPI, gather_times, shard_instances
    repeats = max(1, min(50, -(-400 // max(steps, 1))))
    names = [None] * world
    me = f"rank {rank}: cuda:{dev} {torch.cuda.get_device_name(dev)}"
    if dist is not None:
        dist.all_gather_object(names, me)
    else:
        names = [me]
    if which == "c4":
        total = 64
        mine = shard_instances(total, world, rank)                 # instance ids (= map seeds) of this rank
        B = len(mine)
        insts = [synth.make_instance(G, seed=s, resolution=RES, jitter=True) for s in mine]
        pl = NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=RES, num_instances=B, device_id=dev, stream=stream.cuda_stream,
                        overlap=not a.no_overlap)
        for b, it in enumerate(insts):
            pl.set_map(it.risk.numpy(), b); pl.set_goal(it.goal.numpy(), b)
        states = torch.stack([it.start for it in insts]).cuda()

        def region(n):
            sync(); t0 = time.perf_counter()
            pl.solve_n_async_device(n, states.data_ptr()); pl.flush()
            torch.cuda.synchronize(); dt_ = time.perf_counter() - t0
            if dist is not None:
                dist.barrier()
            pl.sync()
            return dt_
        region(max(a.warmup, 1))
        walls = [region(steps) for _ in range(repeats)]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(stream); pl.solve_n_async_device(max(steps, 100), states.data_ptr()); e1.record(stream)
        pl.flush(); torch.cuda.synchronize(); pl.sync()
        kernel_ms = e0.elapsed_time(e1) / max(steps, 100)
        alg = pl.algorithmic_bytes(injected_noise=False) * B
        units_per_step, scaling = total, "strong"          # 64 instances in total whatever N: the total work is fixed
        workload = (f"BASELINE configs[3]: {total} independent 256x256 instances (map seeds 0..{total - 1}), K={K}, T={T}, sharded "
                    f"{total}/{world} per GPU, one launch of this rank's instances per step, dependent warm-started solves")
        kernel = "bn::rollout_kernel (role kernel)" if B * 17 <= 4352 else "bn::rollout_wave_kernel"
        par = f"instance sharding x{world}, no data-path collective"
        pl.close()
    else:
        K5, T5, G5 = 16384, 100, 512
        inst = synth.make_instance(G5, seed=0, resolution=RES)
        sh = ShardedMPPI(horizon=T5, num_samples=K5, grid_size=G5, resolution=RES, device_id=dev, stream=stream.cuda_stream)
        sh.planner.set_map(inst.risk.numpy()); sh.planner.set_goal(inst.goal.numpy())
        st = inst.start.cuda()

        def region(n):
            sync(); t0 = time.perf_counter()
            for _ in range(n):
                sh.solve(st)
            torch.cuda.synchronize(); dt_ = time.perf_counter() - t0
            if dist is not None:
                dist.barrier()
            sh.planner.sync()
            return dt_
        region(max(a.warmup, 1))
        walls = [region(steps) for _ in range(repeats)]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(stream)
        for _ in range(max(steps, 50)):
            sh.solve(st)
        e1.record(stream); torch.cuda.synchronize(); sh.planner.sync()
        kernel_ms = e0.elapsed_time(e1) / max(steps, 50)
        # this rank's share of the algorithmic bytes: its rollouts' trajectories and weights, the map, + the exchanged partials
        alg = sh.planner.algorithmic_bytes(injected_noise=False) + 2 * (K5 // 64) * (2 + 2 * T5) * 4
        units_per_step, scaling = 1, "strong"
        workload = (f"BASELINE configs[4]: ONE solve K={K5}, T={T5}, {G5}x{G5} map sharded by rollouts ({K5 // world} per GPU): rollouts, "
                    f"all-gather of the softmin partials ({(K5 // 64) * (2 + 2 * T5) * 4} B in total), merge + tail on every rank")
        kernel = "bn::rollout_kernel (role kernel, shard) + all_gather_into_tensor + bn::finish_kernel"
        par = f"rollout sharding x{world}, one RCCL all-gather of the softmin partials per solve" if dist is not None else "one rank, no process group"
        sh.close()
    per_rank = gather_times(walls, coll_dev)
    job = per_rank.max(dim=0).values
    med = float(job.median())
    r_med = int((job - med).abs().argmin())
    value = units_per_step * steps / med
    if rank != 0:
        return None
    out = {"metric": "MPPI solve-steps/sec", "value": value, "unit": "solves/s", "n_gpus": world, "steps": steps, "warmup": a.warmup,
           "ms_per_step": med / steps * 1e3, "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32",
           "data": "synthetic", "repeats": repeats, "world_size": world, "devices": names,
           "per_rank_seconds": [float(x) for x in per_rank[:, r_med]], "host_cpu": host_cpu(),
           "collective_backend": (dist.get_backend() if dist is not None else None),
           "config": {"workload": workload, "parallelism": par, "noise": "philox in-kernel"},
           "roofline": {"bound": "hbm", "achieved": alg / (kernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": alg / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "kernel": kernel, "kernel_ms": kernel_ms,
                        "algorithmic_bytes_per_launch": alg, "note": "rank 0's launch; per-step time between HIP events on the launch stream"}}
    if shared_gpu or (dist is not None and backend != "nccl"):
        out["rehearsal"] = f"ranks shared {ndev} GPU(s), backend {backend}: not a scaling measurement"
    return out


def parity_census():
    """What the default arithmetic costs in parity and what BN_FLAG_REFERENCE_ORDER buys back: counts recorded when the census
    fixtures were captured (tests/golden/make_golden.py census: the imported reference against the three arithmetic modes on the
    same noise; the HIP kernels are bit-exact with modes 1 and 2, tests/test_gpu_census.py).  Data only -- nothing is computed here."""
    path = os.path.join(ROOT, "tests", "golden", "census_summary.json")
    try:
        s = json.load(open(path))
    except Exception as e:
        return {"error": f"{path}: {e}"}

    def rate(recs, trig):
        n = sum(r["rollouts"] for r in recs)
        k = sum(r["over_1e4"][trig] for r in recs)
        return {"rollouts": n, "beyond_1e-4": k, "rate": k / n, "max_dev": max(r["max_dev"][trig] for r in recs)}
    c2 = [s["stored"]["census_c2"], s["unstored_sweep"]["c2_wide"]]
    c5 = [s["stored"]["census_c5"], s["unstored_sweep"]["c5_wide"]]
    return {"tolerance": "max|dX| <= 1e-4 per rollout against the reference (SURVEY 8a (i)); every rollout beyond it is a cell flip "
                         "(tests/helpers.py census_classify)",
            "configs[1] K=1024 T=50 G=256": {"default (spec)": rate(c2, "1"), "BN_FLAG_REFERENCE_ORDER": rate(c2, "2"), "libm in reference order": rate(c2, "0")},
            "configs[4] K=16384 T=100 G=512": {"default (spec)": rate(c5, "1"), "BN_FLAG_REFERENCE_ORDER": rate(c5, "2"), "libm in reference order": rate(c5, "0")},
            "configs[2] K=8192 T=50 sampled slip": {"default (spec)": s["stored"]["census_c3"]["over_1e4"]["1"], "BN_FLAG_REFERENCE_ORDER": s["stored"]["census_c3"]["over_1e4"]["2"],
                                                    "rollouts": s["stored"]["census_c3"]["rollouts"]},
            "source": "tests/golden/census_summary.json (torch " + str(s.get("torch_version")) + ")"}


def rehearse(rank, world, local, backend):
    """Launcher check (CPU test): the ranks meet, exchange who they are and one timing vector, rank 0 reports.
    No planner is created and no `value` is printed: this is not a measurement and not a CPU path of the product."""
    import torch
    import torch.distributed as dist
    from benchnav_amd.sharding import gather_times
    if world > 1:
        dist.init_process_group(backend if backend != "nccl" or torch.cuda.is_available() else "gloo")
    who = [None] * world
    me = {"rank": rank, "local_rank": local, "pid": os.getpid()}
    if world > 1:
        dist.all_gather_object(who, me)
    else:
        who = [me]
    per_rank = gather_times([0.001 * (rank + 1), 0.002 * (rank + 1)], None)
    if world > 1:
        dist.barrier()
        used = dist.get_backend()
        dist.destroy_process_group()
    else:
        used = None
    if rank == 0:
        emit({"rehearsal": True, "n_gpus": world, "world_size": world, "backend": used, "ranks": who,
                          "gathered_shape": list(per_rank.shape), "slowest_per_repeat": [float(x) for x in per_rank.max(dim=0).values]})


def cpu_baseline(inst, seconds, default_threads):
    """The reference's CPU path, as ported in oracle/torch_port.py, on this box's host cores.
    The path is dispatch-bound (~19k ATen calls per solve), so it is timed with 1 thread and with
    torch's default thread count and the faster of the two is reported."""
    import numpy as np
    import torch
    from oracle import torch_port as TP
    pb = TP.Problem(risk=inst.risk, goal=inst.goal, grid_size=G, resolution=RES, x_limits=(0.0, G * RES),
                    y_limits=(0.0, G * RES), sigmas=torch.tensor([0.5, 0.5]), lambda_=0.5, stuck_threshold=0.3,
                    u_min=torch.tensor([0.0, -1.0]), u_max=torch.tensor([1.0, 1.0]))
    runs = {}
    for threads in sorted({1, default_threads}):
        torch.set_num_threads(threads)
        torch.manual_seed(42)
        mean = torch.zeros(T, 2)
        for _ in range(3):
            mean = TP.solve(pb, inst.start, mean, K=K)["Ustar"]
        n, t0 = 0, time.perf_counter()
        while True:
            mean = TP.solve(pb, inst.start, mean, K=K)["Ustar"]      # warm-started chain, noise drawn per solve
            n += 1
            el = time.perf_counter() - t0
            if el >= seconds / 2 or n >= 2000:
                break
        runs[threads] = (n / el, n, el)
    torch.set_num_threads(default_threads)
    best = max(runs, key=lambda k_: runs[k_][0])
    rate, n, el = runs[best]
    cpu = host_cpu()
    out = {"value": rate, "unit": "solves/s", "cores": best, "kind": "port",
           "sample": f"{n} warm-started solves of the same workload (K={K}, T={T}, {G}x{G} map) in {el:.1f} s, "
                     f"PyTorch-CPU port of mppi.py:130-219 (oracle/torch_port.py), torch {torch.__version__}, "
                     f"{cpu['model']}, {cpu['logical_cpus']} logical cpus",
           "by_threads": {str(k_): v[0] for k_, v in runs.items()}}
    # the scalar C oracle on one core, for scale (a stronger CPU implementation than the reference's)
    try:
        from oracle import oracle as O
        p = O.make_params(K, T, G, RES, inst.goal.numpy(), trig=O.TRIG_SPEC)
        eps = np.random.default_rng(0).standard_normal((K, T, 2)).astype(np.float32)
        R, st, mn = inst.risk.numpy(), inst.start.numpy(), np.zeros((T, 2), np.float32)
        O.solve(p, R, st, mn, eps)
        m, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < min(3.0, seconds):
            O.solve(p, R, st, mn, eps)
            m += 1
        out["c_oracle_1core_solves_per_s"] = m / (time.perf_counter() - t1)
    except Exception as e:  # the C oracle is optional for the baseline leg
        out["c_oracle_error"] = str(e)
    return out


if __name__ == "__main__":
    main()
