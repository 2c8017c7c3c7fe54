"""Round-2 launch-time table: microseconds per launch (HIP events, groups of 10 launches) over instances per launch,
kernel (role / wave), lean mode and the grid's XCD interleaving.  K=1024, T=50, 256x256 maps, Philox noise.

    python tools/r2_measure.py                       # the default sweep
    BN_BS=1,64 BN_KERNELS=role BN_LEAN=0,1 BN_PACK=0,1 python tools/r2_measure.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

if os.environ.get("BN_VARIANT"):                              # a library built with other -D flags: tools/_ablate/lib_<variant>.so
    from benchnav_amd import build as _b
    _b.LIB_PATH = os.path.join(ROOT, "tools", "_ablate", "lib_%s.so" % os.environ["BN_VARIANT"])
from benchnav_amd import NativeMPPI, synth

K, T, G = int(os.environ.get("BN_K", 1024)), int(os.environ.get("BN_T", 50)), int(os.environ.get("BN_G", 256))
BS = [int(x) for x in os.environ.get("BN_BS", "1,8,32,60,64,120,128,256").split(",")]
KERNELS = os.environ.get("BN_KERNELS", "role,wave").split(",")
LEANS = [int(x) for x in os.environ.get("BN_LEAN", "0,1").split(",")]
PACKS = os.environ.get("BN_PACK", "auto").split(",")          # auto | 0 | 1
FAIRS = os.environ.get("BN_FAIRS", "auto").split(",")         # auto | 0..3 (bit 0 rotate rollout priorities, bit 1 aux at priority 3)
N = int(os.environ.get("BN_N", 300))
torch.set_num_threads(1)
insts = [synth.make_instance(G, seed=s, jitter=True) for s in range(min(max(BS), 64))]

print(f"# K={K} T={T} G={G}; us per launch (events), M solves/s, TB/s on algorithmic bytes", flush=True)
for B in BS:
    for kern in KERNELS:
        for lean in LEANS:
            for pack, fair in [(p_, f_) for p_ in PACKS for f_ in FAIRS]:
                if fair == "auto":
                    os.environ.pop("BN_FAIR", None)
                else:
                    os.environ["BN_FAIR"] = fair
                if pack == "auto":
                    os.environ.pop("BN_XCD_PACK", None)
                else:
                    os.environ["BN_XCD_PACK"] = pack
                shared = B > len(insts)
                pl = NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, num_instances=B, shared_map=shared,
                                profile=True, stream=0, kernel=kern, lean=bool(lean))
                if shared:
                    pl.set_map(insts[0].risk.numpy()); pl.set_goal(insts[0].goal.numpy())
                    st = torch.stack([insts[0].start] * B).cuda()
                else:
                    for b in range(B):
                        pl.set_map(insts[b].risk.numpy(), b); pl.set_goal(insts[b].goal.numpy(), b)
                    st = torch.stack([insts[b].start for b in range(B)]).cuda()
                torch.cuda.synchronize()
                pl.solve_n_async_device(60, st.data_ptr()); pl.kernel_ms()
                best = 1e9
                for _ in range(3):
                    pl.solve_n_async_device(N, st.data_ptr())
                    best = min(best, pl.kernel_ms()[0] * 1e3)
                by = pl.algorithmic_bytes(injected_noise=False) * B
                pl.close()
                print(f"B={B:4d} {kern:5s} lean={lean} pack={pack:4s} fair={fair:4s}: {best:7.2f} us  {B / best:6.3f} M solves/s  {by / best / 1e6:5.2f} TB/s "
                      f"({by / best / 1e6 / 8.0 * 100:4.1f} % of 8 TB/s)", flush=True)
