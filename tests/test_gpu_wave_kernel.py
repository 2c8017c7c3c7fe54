"""GPU: the one-wave-per-64-rollouts throughput kernel against the five-wave role kernel and the oracle.
Same device functions in the same order per rollout: every output must be bit-identical."""
import numpy as np
import pytest

from helpers import assert_oracle_parity, native_outputs, oracle_metrics

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("K,T,B,noise", [(1024, 50, 3, "philox"), (1000, 33, 2, "kt2"), (2048, 50, 1, "t2k"), (130, 7, 5, "philox"), (64, 1, 1, "kt2")],
                         ids=["c2-B3", "ragged-kt2", "K2048-t2k", "small-B5", "T1"])
def test_wave_kernel_chain_equals_role_kernel_chain(K, T, B, noise):
    import torch
    from benchnav_amd import NativeMPPI, _capi, synth
    from benchnav_amd.mppi import _DevArray
    G, n = 256, 5
    insts = [synth.make_instance(G, seed=20 + b, jitter=True) for b in range(B)]
    rng = np.random.default_rng(2)
    eps = rng.standard_normal((n, B, K, T, 2)).astype(np.float32)
    st = torch.stack([it.start for it in insts]).cuda()
    if noise == "kt2":
        ed, kind = torch.from_numpy(eps).cuda(), _capi.BN_NOISE_DEVICE_KT2
    elif noise == "t2k":
        ed, kind = torch.from_numpy(np.ascontiguousarray(eps.transpose(0, 1, 3, 4, 2))).cuda(), _capi.BN_NOISE_DEVICE_T2K
    else:
        ed, kind = None, _capi.BN_NOISE_PHILOX
    torch.cuda.synchronize()
    res = {}
    for kern in ("role", "wave"):
        with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, num_instances=B, seed=9, store_controls=True, kernel=kern) as pl:
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
        for j, (a_, b_) in enumerate(zip(res["wave"][b], res["role"][b])):
            assert np.array_equal(a_, b_), (b, j)


@pytest.mark.parametrize("lean", [False, True], ids=["full", "lean"])
@pytest.mark.parametrize("T", [1, 2, 3, 8, 9, 29, 30, 31, 32, 33, 35, 39, 40, 41, 49, 63, 100])
def test_parked_controls_over_the_horizon(T, lean):
    """Round 5: the one-wave kernel keeps the controls of its first 30 steps in the lane's own registers (VGPR index mode), one more
    chunk of eight in an LDS tile, and draws the rest again.  Every boundary of that arrangement -- horizons below, at and just above
    the register block, a partial and an odd LDS-parked chunk, horizons far beyond both -- against the role kernel, which has none
    of it: costs, weights, warm start and (with the trajectory dump) every state must be bit-identical, Philox noise, ragged K."""
    import torch
    from benchnav_amd import NativeMPPI, synth
    K, B, G, n = 130, 3, 64, 3
    insts = [synth.make_instance(G, seed=40 + b, jitter=True) for b in range(B)]
    st = torch.stack([it.start for it in insts]).cuda()
    res = {}
    for kern in ("role", "wave"):
        with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, num_instances=B, seed=11, kernel=kern, lean=lean) as pl:
            for b, it in enumerate(insts):
                pl.set_map(it.risk.numpy(), b); pl.set_goal(it.goal.numpy(), b)
            pl.solve_n_async_device(n, st.data_ptr()); pl.sync()
            res[kern] = [(pl.costs(b), pl.weights(b), pl.get_mean(b), pl.states(b)) for b in range(B)]
    for b in range(B):
        for j, (a_, b_) in enumerate(zip(res["wave"][b], res["role"][b])):
            assert np.array_equal(a_, b_), (T, lean, b, j)


def test_wave_kernel_matches_oracle_and_is_the_default_for_large_batches():
    from oracle import oracle as O
    from benchnav_amd import NativeMPPI, synth
    K, T, G, B = 1024, 50, 256, 96                      # 96 x 17 workgroups > 1536: auto-selects the throughput kernel
    insts = [synth.make_instance(G, seed=s, jitter=True) for s in range(B)]
    rng = np.random.default_rng(5)
    eps = rng.standard_normal((B, K, T, 2)).astype(np.float32)
    mean = np.clip(rng.standard_normal((T, 2)) * 0.2 + [0.6, 0.0], [0, -1], [1, 1]).astype(np.float32)
    states = np.stack([it.start.numpy() for it in insts])
    with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, num_instances=B, store_controls=True) as pl:
        for b, it in enumerate(insts):
            pl.set_map(it.risk.numpy(), b); pl.set_goal(it.goal.numpy(), b); pl.set_mean(mean, b)
        us, xs = pl.solve(states, eps)
        for b in (0, 31, 95):
            p = O.make_params(K, T, G, 0.5, insts[b].goal.numpy(), trig=O.TRIG_SPEC)
            orc = O.solve(p, insts[b].risk.numpy(), states[b], mean, eps[b])
            assert_oracle_parity(oracle_metrics(native_outputs(pl, us, xs, b), orc), ctx=f"instance {b}")


def test_episodes_on_a_wave_eligible_handle_use_the_role_kernel_and_agree():
    """A handle large enough for the throughput kernel still runs device-side episodes (closed loop) with the role
    kernel; logs must equal those of a handle forced to the role kernel, and plain solves afterwards still work."""
    import torch
    from benchnav_amd import NativeMPPI, synth
    K, T, G, B, n = 256, 12, 64, 320, 4                  # 320 x 5 workgroups > 1536
    inst = synth.make_instance(G, seed=3)
    lat_std = synth.slip_std_map(G, seed=3).numpy()
    states0 = np.tile(inst.start.numpy(), (B, 1)) + 0.05 * np.arange(B)[:, None].astype(np.float32) % 3
    logs = {}
    for kern in ("auto", "role"):
        with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, num_instances=B, shared_map=True, seed=2, kernel=kern) as pl:
            pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
            pl.env_attach(inst.risk.numpy() * 0.5, lat_std, seed=7)
            st, rw, done = pl.episode(n, states0)
            us, xs = pl.solve(st[-1])
            logs[kern] = (st, rw, done, us, xs, pl.weights(5))
    for a_, b_ in zip(logs["auto"], logs["role"]):
        assert np.array_equal(a_, b_)
