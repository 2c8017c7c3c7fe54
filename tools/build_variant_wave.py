"""Build tools/_ablate/lib_<name>.so where the one-wave kernel's translation units (rollout_wave.hip, rollout_wave_ref.hip), the
helpers (mppi_kernels.hip: LDS sizes) and the host side are compiled with the extra -D flags; the other objects come from the product's
set (benchnav_amd/lib/obj -- build the product first).  ~1.5 minutes:
    python tools/build_variant_wave.py name -DBN_WAVE_LDS_PAD=8192 ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchnav_amd import build as b
name, flags = sys.argv[1], sys.argv[2:]
out_dir = os.path.join(ROOT, "tools", "_ablate")
os.makedirs(out_dir, exist_ok=True)
compile_flags = [f for f in b.HIPCC_FLAGS if f != "-shared"] + ["-DBN_EXPERIMENTS"]
base_dir = os.path.join(b.LIB_DIR, "obj")
mine = ("rollout_wave.hip", "rollout_wave_ref.hip", "mppi_kernels.hip", "mppi_capi.cpp")
procs = []
for s in mine:
    obj = os.path.join(out_dir, f"{os.path.splitext(s)[0]}_{name}.o")
    procs.append((obj, subprocess.Popen([b.hipcc(), *compile_flags, *b.EXTRA_FLAGS.get(s, []), *flags, "-x", "hip", "-c", os.path.join(b.CSRC, s), "-o", obj])))
objs = []
for obj, pr in procs:
    assert pr.wait() == 0, obj
    objs.append(obj)
others = [os.path.join(base_dir, os.path.splitext(s)[0] + ".o") for s in b.SOURCES if s not in mine]
out = os.path.join(out_dir, f"lib_{name}.so")
subprocess.check_call([b.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, *others, "-o", out])
print("built", out)
