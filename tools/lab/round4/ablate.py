"""Build ablated variants of the library (-DBN_ABLATE=mask) and time the rollout kernel of each.
Run on the GPU box:  python tools/ablate.py   (builds were made beforehand by `python tools/ablate.py build`)"""
import os, sys, subprocess, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MASKS = {0: "full", 1: "-stagecost", 4: "-Xstores", 8: "-Utile/ctrlcost", 16: "-sincos", 32: "-gather",
         1 | 2 | 4 | 8: "chain only", 127: "everything off"}
VAR_DIR = os.path.join(ROOT, "tools", "_ablate")

def build():
    from benchnav_amd import build as b
    os.makedirs(VAR_DIR, exist_ok=True)
    procs = []
    for m in MASKS:
        out = os.path.join(VAR_DIR, f"lib_{m}.so")
        cmd = [b.hipcc(), *b.HIPCC_FLAGS, f"-DBN_ABLATE={m}", "-x", "hip", *[os.path.join(b.CSRC, s) for s in b.SOURCES], "-o", out]
        procs.append((out, subprocess.Popen(cmd)))
        if len(procs) % 4 == 0:
            for o, pr in procs[-4:]:
                pr.wait(); print("built", o, pr.returncode)
    for o, pr in procs:
        pr.wait()

def run_one(mask):
    import ctypes, numpy as np, torch
    from benchnav_amd import _capi, build as b
    b.LIB_PATH = os.path.join(VAR_DIR, f"lib_{mask}.so")
    from benchnav_amd import NativeMPPI, synth
    inst = synth.make_instance(256, seed=0)
    res = []
    B = int(os.environ.get("BN_ABLATE_B", "1"))          # instances per launch: 1 = latency regime, 60 = one full resident round
    for noise in ("philox", "t2k"):
        pl = NativeMPPI(horizon=50, num_samples=1024, grid_size=256, resolution=0.5, profile=True, stream=0, num_instances=B, shared_map=True)
        pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
        st = torch.stack([inst.start] * B).cuda(); eps = torch.randn(B, 50, 2, 1024, device="cuda"); torch.cuda.synchronize()
        for it in range(2):
            for _ in range(300):
                if noise == "philox": pl.solve_async_device(st.data_ptr())
                else: pl.solve_async_device(st.data_ptr(), eps.data_ptr(), _capi.BN_NOISE_DEVICE_T2K)
            r = pl.kernel_ms()
        res.append(r[0] * 1e3)
        pl.close()
    print(f"mask {mask:3d} {MASKS[mask]:18s} rollout us: philox {res[0]:.2f}  t2k {res[1]:.2f}", flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    elif len(sys.argv) > 2 and sys.argv[1] == "one":
        run_one(int(sys.argv[2]))
    else:
        for m in MASKS:
            subprocess.call([sys.executable, os.path.abspath(__file__), "one", str(m)])
