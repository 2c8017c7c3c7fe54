"""Wide one-off run of tests/differential.py's other sweeps -- device-side episodes, sampled slip, pairs of planners alive at once:
    python tools/fuzz_features2.py 0 1500 [episode,sampled,pair,ops]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from differential import episode, sampled, pair, ops

if __name__ == "__main__":
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    which = sys.argv[3].split(",") if len(sys.argv) > 3 else ["episode", "sampled", "pair", "ops"]
    for name in which:
        fn = globals()[name]
        tally = {}
        for seed in range(lo, hi):
            try:
                r = fn(seed)
            except Exception as e:                              # noqa: BLE001
                r = "ERROR " + repr(e)[:160]
            key = r.split(":")[0].split(" (")[0]
            tally[key] = tally.get(key, 0) + 1
            if not (r.startswith("ok") or r.startswith("skip")) or "recoveries" in r:
                print(name, seed, r, flush=True)
        print(name, "tally:", tally, flush=True)
