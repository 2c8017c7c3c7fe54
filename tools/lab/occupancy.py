"""Resident rollout workgroups per CU as the HIP runtime computes it (timing build)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes as C, torch
from benchnav_amd import build as b
b.LIB_PATH = os.path.join(ROOT, "tools", "_ablate", "lib_timing.so")
from benchnav_amd import NativeMPPI
pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, stream=0)
pl._lib.bn_mppi_debug_blocks_per_cu.argtypes = [C.c_void_p]
print("blocks per CU:", pl._lib.bn_mppi_debug_blocks_per_cu(pl._h))
prop = torch.cuda.get_device_properties(0)
print("CUs", prop.multi_processor_count, "shared mem per block", getattr(prop, "shared_memory_per_block", None), "per SM", getattr(prop, "shared_memory_per_multiprocessor", None), "regs/SM", getattr(prop, "regs_per_multiprocessor", None), "max threads/SM", prop.max_threads_per_multi_processor)
