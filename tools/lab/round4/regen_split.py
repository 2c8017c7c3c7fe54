"""256 instances on the one-wave kernel: us per launch over the number of steps whose controls the epilogue draws again
(BN_REGEN_STEPS, experiments build: python tools/build_variant_fast.py exp; BN_TOOL_LIB=exp), full and lean mode."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import torch
    from benchnav_amd import build as _b
    if os.environ.get("BN_TOOL_LIB", "main") != "main":
        _b.LIB_PATH = os.path.join(ROOT, "tools", "_ablate", "lib_%s.so" % os.environ["BN_TOOL_LIB"])
    from benchnav_amd import NativeMPPI, synth
    B = int(os.environ.get("BN_B", "256"))
    inst = synth.make_instance(256, seed=0)
    out = []
    for lean in (False, True):
        pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, num_instances=B, shared_map=True, lean=lean)
        pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
        st = torch.stack([inst.start] * B).cuda(); torch.cuda.synchronize()
        pl.solve_n_async_device(30, st.data_ptr()); pl.sync()
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            pl.solve_n_async_device(200, st.data_ptr()); pl.sync()
            best = min(best, (time.perf_counter() - t0) / 200)
        out.append(f"{'lean' if lean else 'full'} {best * 1e6:6.2f} us ({B / best / 1e6:.2f} M solves/s)")
        pl.close()
    print(f"regen_steps={sys.argv[1]:>3s}: " + "   ".join(out), flush=True)
else:
    for r in os.environ.get("BN_SPLITS", "0,8,16,20,24,28,32,36,40,52").split(","):
        env = dict(os.environ, BN_REGEN_STEPS=r)
        subprocess.call([sys.executable, os.path.abspath(__file__), r], env=env)
