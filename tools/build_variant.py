"""Build tools/_ablate/lib_<name>.so with extra -D flags:  python tools/build_variant.py name -DBN_TIMING -DFOO=1 ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchnav_amd import build as b
name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "tools", "_ablate", f"lib_{name}.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call([b.hipcc(), *b.HIPCC_FLAGS, "-DBN_EXPERIMENTS", *flags, "-x", "hip", *[os.path.join(b.CSRC, s) for s in b.SOURCES], "-o", out])
print("built", out)
