"""Static check of rollout_wave_kernel<.., REF = true> ISA for the register-allocation fault described in DESIGN.md 4.15: in every
reference-order instantiation, the packed add that advances the position must read a register pair that the clamps (v_med3_f32)
of the previous step -- or a move from them -- have written.  Compiled WITH the SLP vectoriser the injected-noise variants fail.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math [-fno-slp-vectorize] -S --cuda-device-only \
          -o /tmp/wave_ref.s benchnav_amd/csrc/rollout_wave_ref.hip && python tools/check_wave_isa.py /tmp/wave_ref.s"""
import re
import sys


def kernels(path):
    cur, out = None, {}
    for line in open(path):
        m = re.match(r"^(_ZN2bn\S*rollout_wave_kernel\S*):", line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is not None:
            if line.strip().startswith(";"):
                continue
            out[cur].append(line.rstrip())
            if "s_endpgm" in line:
                cur = None
    return out


bad = 0
for name, ls in kernels(sys.argv[1]).items():
    if not name.endswith("Lb1EEEvNS_11SolveParamsE"):      # REF = true only
        continue
    dests = {int(m.group(1)) for l in ls for m in [re.match(r"\s*v_med3_f32 v(\d+),", l)] if m}
    movs = {}
    for l in ls:
        m = re.match(r"\s*v_mov_b32_e32 v(\d+), v(\d+)", l)
        if m:
            movs.setdefault(int(m.group(1)), set()).add(int(m.group(2)))
    for i, l in enumerate(ls):
        m = re.match(r"\s*v_pk_add_f32 v\[(\d+):\d+\], v\[(\d+):\d+\], v\[(\d+):\d+\]$", l)
        if m and any("v_med3_f32" in x and ("v%s," % m.group(1)) in x for x in ls[i + 1:i + 6]):
            y = int(m.group(2))
            ok = y in dests or any(s in dests for s in movs.get(y, ()))
            print(name[-44:], "position add reads v[%d:%d]" % (y, y + 1), "ok" if ok else "NEVER WRITTEN")
            bad += not ok
print("broken sites:", bad)
sys.exit(1 if bad else 0)
