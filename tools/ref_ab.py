"""A/B of libraries in the REFERENCE-ORDER arithmetic (and the default next to it): one child process per library and round, alternating;
SHA-256 over a 7-solve chain's outputs and the slope of the region time between 50 and 200 dependent solves.
    python tools/ref_ab.py main old [--rounds 3]"""
import hashlib, os, statistics, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(name):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from benchnav_amd import build as _b
    if name != "main":
        _b.LIB_PATH = os.path.join(ROOT, "tools", "_ablate", "lib_%s.so" % name)
    from benchnav_amd import NativeMPPI, synth
    torch.set_num_threads(1)
    inst = synth.make_instance(256, seed=0)
    st = inst.start.cuda()
    stream = torch.cuda.Stream()
    out = []
    for ref in (True, False):
        h = hashlib.sha256()
        pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, stream=stream.cuda_stream, reference_order=ref)
        pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
        pl.solve_n_async_device(7, st.data_ptr()); pl.sync()
        for a in (pl.states(), pl.costs(), pl.weights(), pl.get_mean()):
            h.update(np.ascontiguousarray(a).tobytes())
        pl.solve_n_async_device(300, st.data_ptr()); pl.sync()
        row = {}
        for K in (50, 200):
            ts = []
            for _ in range(60):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                pl.solve_n_async_device(K, st.data_ptr()); pl.flush(); torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            row[K] = statistics.median(ts) * 1e6
        pl.close()
        out.append("%s sha %s slope %6.3f us/solve" % ("reference order" if ref else "default", h.hexdigest()[:12], (row[200] - row[50]) / 150))
    print("%-8s %s" % (name, " | ".join(out)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
        sys.exit(0)
    args = sys.argv[1:]
    rounds = 3
    if "--rounds" in args:
        i = args.index("--rounds"); rounds = int(args[i + 1]); del args[i:i + 2]
    for r in range(rounds):
        for name in args:
            subprocess.call([sys.executable, os.path.abspath(__file__), "--child", name])
