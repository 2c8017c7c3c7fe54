"""Build tools/_ablate/lib_<name>.so where only the latency / role kernel of the Philox noise source (rollout_role_philox.hip) is
compiled with the extra -D flags; every other object comes from a base set: the product's (benchnav_amd/lib/obj) or, with
`--timing`, a -DBN_TIMING set built once into tools/_ablate/obj_timing (so that the stamp tools work on variants).  ~1 minute:
    python tools/build_variant_fast.py [--timing] name -DBN_ABLATE=16 ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchnav_amd import build as b
args = sys.argv[1:]
timing = "--timing" in args
plain = "--plain" in args          # without -DBN_EXPERIMENTS: the kernels as the shipped library compiles them (that define alone moves the
                                   # period of dependent solves by 0.8 us, DESIGN_NOTEBOOK.md R5.4); no environment switches then
args = [a for a in args if a not in ("--timing", "--plain")]
name, flags = args[0], args[1:]
SRC = os.environ.get("BN_VARIANT_SRC", "rollout_role_philox.hip")      # the one kernel source compiled with the extra flags
out_dir = os.path.join(ROOT, "tools", "_ablate")
os.makedirs(out_dir, exist_ok=True)
compile_flags = [f for f in b.HIPCC_FLAGS if f != "-shared"] + ([] if plain else ["-DBN_EXPERIMENTS"])
base_dir = os.path.join(b.LIB_DIR, "obj")
if timing:
    flags = ["-DBN_TIMING"] + flags
    base_dir = os.path.join(out_dir, "obj_timing")
    os.makedirs(base_dir, exist_ok=True)
    stale = [s for s in b.SOURCES if s != SRC and
             (not os.path.exists(os.path.join(base_dir, os.path.splitext(s)[0] + ".o")) or
              os.path.getmtime(os.path.join(base_dir, os.path.splitext(s)[0] + ".o")) < max(os.path.getmtime(os.path.join(b.CSRC, f)) for f in os.listdir(b.CSRC)))]
    procs = [subprocess.Popen([b.hipcc(), *compile_flags, "-DBN_TIMING", "-x", "hip", "-c", os.path.join(b.CSRC, s), "-o",
                               os.path.join(base_dir, os.path.splitext(s)[0] + ".o")]) for s in stale]
    for pr in procs:
        assert pr.wait() == 0
obj = os.path.join(out_dir, f"{os.path.splitext(SRC)[0]}_{name}.o")
subprocess.check_call([b.hipcc(), *compile_flags, *b.EXTRA_FLAGS.get(SRC, []), *flags, "-x", "hip", "-c", os.path.join(b.CSRC, SRC), "-o", obj])
# the host side as well: its experiment switches (environment variables) exist only with -DBN_EXPERIMENTS
capi = os.path.join(out_dir, f"capi_{name}.o")
subprocess.check_call([b.hipcc(), *compile_flags, *flags, "-x", "hip", "-c", os.path.join(b.CSRC, "mppi_capi.cpp"), "-o", capi])
others = [capi] + [os.path.join(base_dir, os.path.splitext(s)[0] + ".o") for s in b.SOURCES if s not in (SRC, "mppi_capi.cpp")]
out = os.path.join(out_dir, f"lib_{name}.so")
subprocess.check_call([b.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", obj, *others, "-o", out])
print("built", out)
