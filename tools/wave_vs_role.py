"""Role-split latency kernel vs one-wave throughput kernel over the number of instances per launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
if os.environ.get("BN_VARIANT"):
    from benchnav_amd import build as b
    b.LIB_PATH = os.path.join(ROOT, "tools", "_ablate", "lib_%s.so" % os.environ["BN_VARIANT"])
from benchnav_amd import NativeMPPI, synth
inst = synth.make_instance(256, seed=0)
for B in [int(x) for x in os.environ.get("BN_BS", "1,8,16,32,48,60,64,96,128,192,256,384").split(",")]:
    row = []
    for kern in ("role", "wave"):
        pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, num_instances=B, shared_map=True, profile=True,
                        stream=0, kernel=kern)
        pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
        st = torch.stack([inst.start] * B).cuda(); torch.cuda.synchronize()
        pl.solve_n_async_device(60, st.data_ptr()); pl.kernel_ms()
        pl.solve_n_async_device(200, st.data_ptr()); r = pl.kernel_ms()[0] * 1e3
        row.append(r); bytes_ = pl.algorithmic_bytes(injected_noise=False) * B
        pl.close()
    print(f"B={B:4d}: role {row[0]:7.1f} us ({B/row[0]:.2f} M solves/s, {bytes_/row[0]/1e6:5.2f} TB/s)   wave {row[1]:7.1f} us ({B/row[1]:.2f} M solves/s, {bytes_/row[1]/1e6:5.2f} TB/s)", flush=True)
