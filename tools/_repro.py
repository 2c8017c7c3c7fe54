import sys, os, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import fuzz_features as F
seed = int(sys.argv[1])
for i in range(int(sys.argv[2])):
    t0 = time.time(); r = F.run(seed); print(i, r, f"{time.time() - t0:.2f}s", flush=True)
