"""Wide one-off run of tests/differential.py's first sweep (kernel family, overlap, lean, window, call cuts: no result may change):
    python tools/fuzz_features.py 0 6000          FUZZ_BREAK=1: self-test, every case must then report a mismatch"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from differential import case, run

if __name__ == "__main__":
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    tally = {}
    for seed in range(lo, hi):
        try:
            r = run(seed)
        except Exception as e:                                  # noqa: BLE001
            r = "ERROR " + repr(e)[:200]
        key = r.split(":")[0].split(" (")[0]
        tally[key] = tally.get(key, 0) + 1
        if not r.startswith("ok") or "recoveries" in r:
            c = case(seed)
            print(seed, r, {k: c[k] for k in ("B", "K", "T", "G", "noise", "n", "knobs", "cuts")}, {k: c["common"][k] for k in ("resolution", "dt", "u_max", "reference_order", "shared_map")}, flush=True)
    print("tally:", tally)
