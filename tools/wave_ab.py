"""A/B of one-wave-kernel variants (tools/build_variant_wave.py): for every library named (`main` = the product's) one child process
per round, alternating.  A child prints a SHA-256 over the outputs (trajectories, costs, weights, U*) of 3 dependent solves of 20
instances on the one-wave kernel, full and lean, and the time per launch of B instances (BN_BS, default 256,64), full and lean.
    python tools/wave_ab.py main trim1 [--rounds 2]"""
import hashlib, os, subprocess, sys, time
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
    ref = os.environ.get("BN_REF", "0") == "1"
    ov = os.environ.get("BN_OVERLAP", "1") == "1"      # BN_OVERLAP=0: every launch on one stream (what rocprofv3 times per kernel)
    digests = []
    for lean in (False, True):
        h = hashlib.sha256()
        B = 20
        insts = [synth.make_instance(256, seed=b) for b in range(B)]
        st = torch.stack([it.start for it in insts]).cuda()
        pl = NativeMPPI(horizon=50, num_samples=1000, grid_size=256, resolution=0.5, num_instances=B, lean=lean, kernel="wave", reference_order=ref, overlap=ov)
        for b, it in enumerate(insts):
            pl.set_map(it.risk.numpy(), b); pl.set_goal(it.goal.numpy(), b)
        pl.solve_n_async_device(3, st.data_ptr()); pl.sync()
        outs = [pl.costs(b) for b in range(B)] + [pl.weights(b) for b in range(B)] + [pl.get_mean(b) for b in range(B)]
        if not lean:
            outs += [pl.states(b) for b in (0, B - 1)]
        for a in outs:
            h.update(np.ascontiguousarray(a).tobytes())
        digests.append(h.hexdigest()[:12])
        pl.close()
    line = "%-10s sha full %s lean %s |" % (name, digests[0], digests[1])
    for B in [int(x) for x in os.environ.get("BN_BS", "256,64").split(",")]:
        insts = [synth.make_instance(256, seed=b) for b in range(B)]
        st = torch.stack([it.start for it in insts]).cuda()
        for lean in (False, True):
            pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, num_instances=B, lean=lean, kernel="wave", reference_order=ref, overlap=ov)
            for b, it in enumerate(insts):
                pl.set_map(it.risk.numpy(), b); pl.set_goal(it.goal.numpy(), b)
            pl.solve_n_async_device(50, st.data_ptr()); pl.sync()
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                pl.solve_n_async_device(300, st.data_ptr()); pl.sync()
                best = min(best, (time.perf_counter() - t0) / 300)
            line += " B=%d %s %6.2f us" % (B, "lean" if lean else "full", best * 1e6)
            pl.close()
    print(line, flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
        sys.exit(0)
    args = sys.argv[1:]
    rounds = 2
    if "--rounds" in args:
        i = args.index("--rounds"); rounds = int(args[i + 1]); del args[i:i + 2]
    for r in range(rounds):
        for name in args:
            subprocess.call([sys.executable, os.path.abspath(__file__), "--child", name])
