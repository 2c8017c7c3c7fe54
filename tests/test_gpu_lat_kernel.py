"""GPU: the barrier-free latency variant of the rollout kernel (rollout_lat_kernel: all noise drawn up front, full-length
ring, progress counter instead of per-chunk barriers) against the role kernel.  Same device functions in the same order per
rollout: every output must be bit-identical, for every noise source, ragged K and T, lean mode, over warm-started chains."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("K,T,B,noise,lean", [(1024, 50, 1, "philox", False), (1024, 50, 3, "philox", True), (1000, 33, 2, "kt2", False),
                                              (2048, 50, 1, "t2k", False), (130, 7, 5, "philox", False), (64, 1, 1, "kt2", False),
                                              (320, 3, 2, "t2k", True), (192, 100, 1, "philox", False), (256, 9, 1, "philox", False)],
                         ids=["c2", "c2-B3-lean", "ragged-kt2", "K2048-t2k", "small-B5", "T1", "T3-lean", "T100", "T9"])
def test_lat_kernel_chain_equals_role_kernel_chain(K, T, B, noise, lean):
    import torch
    from benchnav_amd import NativeMPPI, _capi, synth
    from benchnav_amd.mppi import _DevArray
    G, n = 256, 5
    insts = [synth.make_instance(G, seed=30 + b, jitter=True) for b in range(B)]
    eps = np.random.default_rng(4).standard_normal((n, B, K, T, 2)).astype(np.float32)
    st = torch.stack([it.start for it in insts]).cuda()
    if noise == "kt2":
        ed, kind = torch.from_numpy(eps).cuda(), _capi.BN_NOISE_DEVICE_KT2
    elif noise == "t2k":
        ed, kind = torch.from_numpy(np.ascontiguousarray(eps.transpose(0, 1, 3, 4, 2))).cuda(), _capi.BN_NOISE_DEVICE_T2K
    else:
        ed, kind = None, _capi.BN_NOISE_PHILOX
    torch.cuda.synchronize()
    res = {}
    for kern in ("role", "lat"):
        with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, num_instances=B, seed=9, store_controls=True, kernel=kern, lean=lean) as pl:
            for b, it in enumerate(insts):
                pl.set_map(it.risk.numpy(), b); pl.set_goal(it.goal.numpy(), b)
            if ed is None:
                pl.solve_n_async_device(n, st.data_ptr())
            else:
                pl.solve_n_async_device(n, st.data_ptr(), ed.data_ptr(), kind, n, eps[0].size)
            pl.sync()
            xs = torch.as_tensor(_DevArray(pl.device_buffer(_capi.BN_BUF_XSTAR)[0], (B, T + 1, 3)), device="cuda").cpu().numpy()
            res[kern] = [(pl.states(b), pl.controls(b), pl.costs(b), pl.weights(b), pl.get_mean(b), xs[b].copy()) for b in range(B)]
    for b in range(B):
        for j, (a_, b_) in enumerate(zip(res["lat"][b], res["role"][b])):
            assert np.array_equal(a_, b_), (b, j)


@pytest.mark.parametrize("K,T,B", [(4096, 20, 1), (3000, 12, 2), (2100, 50, 1)], ids=["K4096", "K3000-B2", "K2100"])
def test_mid_size_solves_merge_in_the_prologue_like_the_two_launch_path(K, T, B):
    """33 to 64 workgroups per instance: every rollout workgroup re-merges the previous solve's partials itself (pipelined mode
    and the latency kernel, rows beyond the first 16 loaded in groups) instead of the ticket merge.  Independent check: the
    two-launch path (BN_FLAG_NO_PIPELINE: rollouts, then the tail kernel merges) over a chain of warm-started solves, overlapped
    and on one stream."""
    import torch
    from benchnav_amd import NativeMPPI, synth
    G, n = 256, 5
    insts = [synth.make_instance(G, seed=50 + b, jitter=True) for b in range(B)]
    st = torch.stack([it.start for it in insts]).cuda()
    torch.cuda.synchronize()
    res = {}
    for name, kw in (("two-launch", dict(pipeline=False)), ("lat-overlap", dict(kernel="lat")), ("lat-one-stream", dict(kernel="lat", overlap=False)),
                     ("role-overlap", dict(kernel="role"))):
        with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, num_instances=B, seed=3, **kw) as pl:
            for b, it in enumerate(insts):
                pl.set_map(it.risk.numpy(), b); pl.set_goal(it.goal.numpy(), b)
            pl.solve_n_async_device(n, st.data_ptr())
            pl.sync()
            res[name] = [(pl.states(b), pl.costs(b), pl.weights(b), pl.get_mean(b)) for b in range(B)]
    for name in ("lat-overlap", "lat-one-stream", "role-overlap"):
        for b in range(B):
            for j, (a_, b_) in enumerate(zip(res[name][b], res["two-launch"][b])):
                assert np.array_equal(a_, b_), (name, b, j)


def test_lat_kernel_matches_the_oracle_and_the_reference_fixture():
    from helpers import (assert_oracle_parity, assert_within, load_case, native_outputs, native_planner_for, oracle_metrics, oracle_params_for,
                         parity_metrics)
    from oracle import oracle as O
    fx = load_case("c2")
    p = oracle_params_for(fx, O.TRIG_SPEC)
    with native_planner_for(fx, kernel="lat") as pl:
        pl.set_map(fx["R"]); pl.set_goal(fx["goal"])
        for i in range(int(fx["n_solves"])):
            pl.set_mean(fx[f"mean_{i}"])
            us, xs = pl.solve(fx[f"state_{i}"], fx[f"eps_{i}"])
            got = native_outputs(pl, us, xs)
            assert_oracle_parity(oracle_metrics(got, O.solve(p, fx["R"], fx[f"state_{i}"], fx[f"mean_{i}"], fx[f"eps_{i}"])), ctx=f"solve {i}")
            assert_within(parity_metrics(got, fx, i), ctx=f"reference solve {i}")
