"""A/B of kernel variants built by tools/build_variant_fast.py: for every library named on the command line (`main` = the product's)
one child process per round, rounds alternating between the libraries (box-to-box and process-to-process spread is larger than the
effects looked for).  A child prints
  * a SHA-256 over the outputs of a 7-solve dependent chain at configs[1] size (trajectory batch, costs, weights, U*, X*) --
    a variant that changes instruction selection and nothing else must print the same digest as `main`;
  * the region times of tools/region_overhead.py at K = 1, 20, 200 (overlapped launches) and the slope between K = 50 and 200.
    python tools/variant_ab.py main asm2 [--rounds 3]"""
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
    h = hashlib.sha256()
    for res, G in ((0.5, 256), (0.3, 256)):
        inst_r = synth.make_instance(G, seed=3, resolution=res)
        pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=G, resolution=res, stream=stream.cuda_stream)
        pl.set_map(inst_r.risk.numpy()); pl.set_goal(inst_r.goal.numpy())
        sr = inst_r.start.cuda()
        pl.solve_n_async_device(7, sr.data_ptr()); pl.sync()
        for a in (pl.states(), pl.costs(), pl.weights(), pl.get_mean()):
            h.update(np.ascontiguousarray(a).tobytes())
        pl.close()
    pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, stream=stream.cuda_stream)
    pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
    pl.solve_n_async_device(300, st.data_ptr()); pl.sync()
    row = {}
    for K in (1, 20, 50, 200):
        ts = []
        for _ in range(80):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            pl.solve_n_async_device(K, st.data_ptr()); pl.flush(); torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        row[K] = statistics.median(ts) * 1e6
    pl.close()
    slope = (row[200] - row[50]) / 150
    print("%-10s sha %s  K=1 %6.1f  K=20 %6.1f  K=200 %7.1f  slope %5.3f us/solve" % (name, h.hexdigest()[:16], row[1], row[20], row[200], slope), flush=True)


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
