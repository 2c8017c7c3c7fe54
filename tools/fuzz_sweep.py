"""One-off wide sweep of the randomised parity cases of tests/test_gpu_fuzz.py over seeds the suite does not hold:
    python tools/fuzz_sweep.py 2000 2600      -> failures (if any) with their seeds; both arithmetics, sampled slip, DWA."""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as F
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = []
names = [("default", lambda s: F._random_configuration(s, False)), ("reference-order", lambda s: F._random_configuration(s, True))]
for extra in ("test_random_sampled_slip_configuration_matches_oracle", "test_random_dwa_configuration_matches_oracle"):
    f = getattr(F, extra)
    names.append((extra, getattr(f, "__wrapped__", f)))
for name, fn in names:
    n = 0
    for seed in range(lo, hi):
        try:
            fn(seed); n += 1
        except Exception as e:                                  # noqa: BLE001 -- report and go on
            bad.append((name, seed, repr(e)[:300]))
    print(f"{name}: {n} of {hi - lo} seeds passed", flush=True)
for b in bad[:40]:
    print("FAIL", b)
print("failures:", len(bad))
