"""MPPI.forward (mppi.py:130-219) as ONE launch: on the latency kernel the solve's tail -- softmin merge, U*, the first-action mailbox,
X*, the weights -- rides in the rollout launch as a second aux workgroup that waits on the device for the rollout workgroups of its own
launch (bn_mppi_forward_async / bn_mppi_forward_state_async / bn_mppi_solve, ABI 4).  Bit-identical to the two-launch path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _xstar_ustar(pl, T, B):
    import torch
    from benchnav_amd import _capi
    from benchnav_amd.mppi import _DevArray
    n = B * (T * 2 + (T + 1) * 3)
    blk = torch.as_tensor(_DevArray(pl.device_buffer(_capi.BN_BUF_USTAR_XSTAR)[0], (n,)), device="cuda").cpu().numpy()
    return blk[:B * T * 2].reshape(B, T, 2).copy(), blk[B * T * 2:].reshape(B, T + 1, 3).copy()


def _snapshot(pl, T, B):
    us, xs = _xstar_ustar(pl, T, B)
    return [(pl.states(b), pl.costs(b), pl.weights(b), pl.get_mean(b), us[b], xs[b]) for b in range(B)]


# K=1024: the headline (granule-polling tail); K=2048: 32 workgroups per instance (counter-waiting tail); B=3: several instances' tails
@pytest.mark.parametrize("K,T,B,ref_order,noise", [(1024, 50, 1, False, "philox"), (1024, 50, 1, True, "philox"), (1024, 50, 1, False, "kt2"),
                                                   (128, 20, 1, False, "philox"), (2048, 30, 1, False, "philox"), (2048, 30, 1, True, "kt2"),
                                                   (1024, 50, 3, False, "philox"), (320, 33, 2, True, "philox"), (100, 7, 1, False, "philox")],
                         ids=["c2", "c2-ref", "c2-kt2", "c1", "K2048", "K2048-ref-kt2", "B3", "ragged-B2-ref", "K100-T7"])
def test_one_launch_forward_equals_two_launch_chain(K, T, B, ref_order, noise):
    """A chain of warm-started forward() calls (every one consumed: first action, caller's output block) on the one-launch path against the
    same chain with BN_FLAG_NO_PIPELINE (rollout kernel + stand-alone tail): every output of every step bit for bit."""
    import torch
    from benchnav_amd import NativeMPPI, _capi, synth
    G, n = 256, 4
    inst = synth.make_instance(G, seed=3)
    rng = np.random.default_rng(7)
    states = [(np.tile(inst.start.numpy(), (B, 1)) + 0.15 * i + 0.1 * np.arange(B)[:, None]).astype(np.float32) for i in range(n)]
    eps = torch.from_numpy(rng.standard_normal((n, B, K, T, 2)).astype(np.float32)).cuda() if noise == "kt2" else None
    dstates = [torch.from_numpy(s).cuda() for s in states]
    torch.cuda.synchronize()
    n_out = B * (T * 2 + (T + 1) * 3)
    res = {}
    for mode in ("one", "one_state", "two"):
        with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, num_instances=B, shared_map=True, seed=9,
                        pipeline=(mode != "two"), reference_order=ref_order) as pl:
            pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
            assert pl.launches_per_forward() == (2 if mode == "two" else 1)
            steps = []
            for i in range(n):
                out = torch.full((n_out,), float("nan"), device="cuda")
                eptr, kind = (eps[i].data_ptr(), _capi.BN_NOISE_DEVICE_KT2) if eps is not None else (None, _capi.BN_NOISE_PHILOX)
                if mode == "one_state":
                    pl.forward_state_async(states[i], eptr, kind, out.data_ptr())
                else:
                    pl.forward_async_device(dstates[i].data_ptr(), eptr, kind, out.data_ptr())
                fa = np.stack([pl.first_action(b) for b in range(B)])
                pl.sync()
                snap = _snapshot(pl, T, B)
                o = out.cpu().numpy()
                for b in range(B):                                   # the caller's block holds what the planner's does; the mailbox U*[0]
                    assert np.array_equal(o[:B * T * 2].reshape(B, T, 2)[b], snap[b][4])
                    assert np.array_equal(o[B * T * 2:].reshape(B, T + 1, 3)[b], snap[b][5])
                    assert np.array_equal(fa[b], snap[b][4][0])
                steps.append(snap)
            res[mode] = steps
    for mode in ("one", "one_state"):
        for i in range(n):
            for b in range(B):
                for j, (got, ref) in enumerate(zip(res[mode][i][b], res["two"][i][b])):
                    assert np.array_equal(got, ref), (mode, i, b, j)
    assert np.isfinite(res["one"][-1][0][5]).all() and np.abs(res["one"][-1][0][3]).max() > 0


def test_synchronous_solve_is_one_launch_and_matches_the_oracle():
    """bn_mppi_solve (host buffers in, host buffers out) takes the one-launch path too: bit-exact trajectories / costs against the oracle,
    U*, X*, weights within the oracle tolerances (tests/helpers.py)."""
    from benchnav_amd import NativeMPPI, synth
    from oracle import oracle as O
    G, K, T, res = 256, 1024, 50, 0.5
    inst = synth.make_instance(G, seed=0, resolution=res)
    R, state, goal = inst.risk.numpy(), inst.start.numpy(), inst.goal.numpy()
    eps = np.random.default_rng(0).standard_normal((K, T, 2)).astype(np.float32)
    with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=res) as pl:
        pl.set_map(R); pl.set_goal(goal)
        assert pl.launches_per_forward() == 1
        us, xs = pl.solve(state, eps)
        X, c, w = pl.states(), pl.costs(), pl.weights()
    orc = O.solve(O.make_params(K, T, G, res, goal, trig=O.TRIG_SPEC), R, state, np.zeros((T, 2), np.float32), eps)
    assert np.array_equal(X, orc["X"]) and np.array_equal(c, orc["cost"])
    assert np.abs(w - orc["w"]).max() < 5e-7 and np.abs(us[0] - orc["Ustar"]).max() < 2e-6 and np.abs(xs[0] - orc["Xstar"]).max() < 1e-5


def test_pending_tail_is_flushed_before_a_one_launch_forward():
    """solve_async leaves its tail pending (it would ride in the next launch); a forward() behind it writes that tail with its own kernel
    first and then takes the one-launch path: same results as the two-launch chain."""
    import torch
    from benchnav_amd import NativeMPPI, synth
    G, K, T = 128, 512, 25
    inst = synth.make_instance(G, seed=4)
    st = torch.from_numpy(inst.start.numpy()).cuda()
    torch.cuda.synchronize()
    res = {}
    for mode in ("one", "two"):
        with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, seed=2, pipeline=(mode == "one")) as pl:
            pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
            pl.solve_async_device(st.data_ptr())
            pl.solve_async_device(st.data_ptr())
            pl.forward_async_device(st.data_ptr())
            fa = pl.first_action()
            pl.solve_n_async_device(5, st.data_ptr())
            pl.forward_state_async(inst.start.numpy())
            fb = pl.first_action()
            pl.sync()
            res[mode] = _snapshot(pl, T, 1)[0] + (fa, fb)
    for j, (a, b) in enumerate(zip(res["one"], res["two"])):
        assert np.array_equal(a, b), j


def test_drop_in_class_takes_a_host_state_by_value():
    """MPPI.forward(state) with a CPU tensor (the reference moves it, mppi.py:140-144): taken by value at the call -- same results as the
    same state on the device, the caller's tensor untouched and free to be overwritten right after the call."""
    import torch
    from helpers import load_case, mppi_for_fixture
    fx = load_case("c2")
    outs = {}
    for where in ("cpu", "cuda"):
        solver = mppi_for_fixture(fx, noise="philox", store_controls=False)
        state = torch.tensor(fx["state_0"], device=where)
        seq = []
        for i in range(3):
            keep = state.clone()
            U, X = solver(state)
            if where == "cpu":
                assert torch.equal(state, keep)
                state = state.clone()                    # a fresh tensor per step, like env.step's return value
                state.add_(0.05)
            else:
                state = state + 0.05
            a = solver.first_action().clone()
            torch.cuda.synchronize()
            assert torch.equal(a, U[0].cpu())
            seq.append((U.cpu().numpy().copy(), X.cpu().numpy().copy(), solver._weights.cpu().numpy().copy()))
        outs[where] = seq
    for (u0, x0, w0), (u1, x1, w1) in zip(outs["cpu"], outs["cuda"]):
        assert np.array_equal(u0, u1) and np.array_equal(x0, x1) and np.array_equal(w0, w1)


def test_forward_entry_points_on_handles_the_latency_kernel_does_not_serve():
    """The role kernel (24 instances), the ticket path (K = 8192) and the one-wave kernel (300 instances): bn_mppi_forward_async and
    bn_mppi_forward_state_async (states staged and uploaded there) keep their two launches and agree with each other and with the
    BN_FLAG_NO_PIPELINE chain, first action and caller's block included."""
    import torch
    from benchnav_amd import NativeMPPI, synth
    G = 128
    inst = synth.make_instance(G, seed=8)
    for K, T, B in ((1024, 30, 24), (8192, 25, 1), (256, 20, 300)):
        states = [(np.tile(inst.start.numpy(), (B, 1)) + 0.1 * i + 0.01 * np.arange(B)[:, None]).astype(np.float32) for i in range(3)]
        dstates = [torch.from_numpy(s).cuda() for s in states]
        torch.cuda.synchronize()
        n_out = B * (T * 2 + (T + 1) * 3)
        res = {}
        for mode in ("device", "host", "two"):
            with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, num_instances=B, shared_map=True, seed=4, pipeline=(mode != "two")) as pl:
                pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
                assert pl.launches_per_forward() == 2 and not pl.host_paced()
                outs = []
                for i in range(3):
                    out = torch.zeros(n_out, device="cuda")
                    torch.cuda.synchronize()
                    if mode == "host":
                        pl.forward_state_async(states[i], None, 0, out.data_ptr())
                    else:
                        pl.forward_async_device(dstates[i].data_ptr(), None, 0, out.data_ptr())
                    fa = pl.first_action(B - 1).copy()
                    pl.sync()
                    o = out.cpu().numpy()
                    assert np.array_equal(fa, o[:B * T * 2].reshape(B, T, 2)[B - 1, 0])
                    outs.append((o, pl.weights(B - 1), pl.costs(0)))
                res[mode] = outs
        for mode in ("device", "host"):
            for i in range(3):
                for j in range(3):
                    assert np.array_equal(res[mode][i][j], res["two"][i][j]), (K, B, mode, i, j)
