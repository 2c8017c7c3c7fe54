import os, sys, time, statistics
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
torch.set_num_threads(1)
torch.cuda.set_device(0)
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29656")
mode = os.environ.get("MODE", "init_first")
if mode == "init_first":
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0), rank=0, world_size=1)
from benchnav_amd import NativeMPPI, synth
inst = synth.make_instance(256, seed=0)
st = inst.start.cuda()
stream = torch.cuda.current_stream()
KK = int(os.environ.get('KK', '200'))
def probe(tag, use_barrier=False):
    pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, stream=stream.cuda_stream)
    pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
    pl.solve_n_async_device(300, st.data_ptr()); pl.sync()
    rows = []
    for rep in range(40):
        if use_barrier: dist.barrier()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pl.solve_n_async_device(KK, st.data_ptr()); t1 = time.perf_counter()
        pl.flush(); torch.cuda.synchronize(); t2 = time.perf_counter()
        if use_barrier: dist.barrier()
        pl.sync()
        rows.append(((t1 - t0) / KK * 1e6, (t2 - t0) / KK * 1e6))
    print(tag, "enqueue us/launch %.2f  total us/solve %.2f" % (statistics.median(r[0] for r in rows), statistics.median(r[1] for r in rows)), flush=True)
    pl.close()
probe(mode + ":")
if mode != "init_first":
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0), rank=0, world_size=1)
probe("with barriers around the regions:", True)
probe("again without:")
