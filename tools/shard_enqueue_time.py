"""Host time per bn_mppi_shard_solve_async against the GPU time per solve (K=16384, T=100, 512x512, one rank over RCCL)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
from benchnav_amd import synth
from benchnav_amd.sharding import ShardedMPPI
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0), rank=0, world_size=1)
K, T, G = int(os.environ.get("BN_K", 16384)), int(os.environ.get("BN_T", 100)), 512
inst = synth.make_instance(G, seed=0)
st = inst.start.cuda()
sh = ShardedMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5)
sh.planner.set_map(inst.risk.numpy()); sh.planner.set_goal(inst.goal.numpy())
for _ in range(50): sh.solve(st)
torch.cuda.synchronize(); sh.planner.sync()
for n in (200, 200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): sh.solve(st)
    t1 = time.perf_counter()
    sh.planner.flush(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"fused={sh._fused}: host enqueue {1e6 * (t1 - t0) / n:.1f} us per solve, total {1e6 * (t2 - t0) / n:.1f} us per solve", flush=True)
sh.close(); dist.destroy_process_group()
