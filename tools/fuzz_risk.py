"""One-off: tests/test_gpu_fuzz.py's risk-map case over more seeds.   python tools/fuzz_risk.py 1000 1300"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as F
f = getattr(F.test_random_risk_map_matches_oracle, "__wrapped__", F.test_random_risk_map_matches_oracle)
bad = []
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    try: f(seed)
    except Exception as e: bad.append((seed, repr(e)[:300]))   # noqa: BLE001
for b in bad[:20]: print("FAIL", b)
print("risk map failures:", len(bad), "of", int(sys.argv[2]) - int(sys.argv[1]))
