"""Regenerate benchnav_amd/csrc/wave_park.h for a register-block base:  python tools/gen_wave_park.py 68   (block = v[base .. 127])"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
base = int(sys.argv[1]); n = (128 - base) & ~3
path = os.path.join(ROOT, "benchnav_amd", "csrc", "wave_park.h")
s = open(path).read()
s = re.sub(r"#define BN_PARK_BASE \d+", f"#define BN_PARK_BASE {base}", s)
s = re.sub(r"#define BN_PARK_CLOBBERS .*", "#define BN_PARK_CLOBBERS " + ", ".join('"v%d"' % (base + i) for i in range(n)), s)
s = re.sub(r"constexpr int kParkSteps = \d+;.*", f"constexpr int kParkSteps = {n // 2};        // {n} registers: v{base} .. v{base + n - 1}", s)
open(path, "w").write(s)
hp = os.path.join(ROOT, "benchnav_amd", "csrc", "mppi_kernels.h")
h = open(hp).read()
h = re.sub(r"constexpr int kWaveParkSteps = \d+;", f"constexpr int kWaveParkSteps = {n // 2};", h)
open(hp, "w").write(h)
print("base", base, "steps", n // 2)
