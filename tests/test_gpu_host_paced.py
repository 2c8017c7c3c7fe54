"""BN_FLAG_HOST_PACED / MPPI(host_loop=True): the loop of test_mppi.py:174-181 -- one forward(state) per control step, the state living on
the host, the host consuming the first action -- with the NEXT solve's launch enqueued one step ahead and waiting on the device for its
state (rollout_lat.inc, HOSTP).  Every output of every step must be bit-identical to the one-launch path's: same Philox positions, same
merges, whatever happens to the waiting launch in between (cancelled, given up, a state far from the previous one)."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _planner(K, T, G, inst, paced, resolution=0.5, **kw):
    from benchnav_amd import NativeMPPI
    pl = NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=resolution, seed=9, stream=0, host_paced=paced, **kw)
    pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
    return pl


def _states(inst, n, jump_at=()):
    rng = np.random.default_rng(3)
    st = inst.start.numpy().astype(np.float32).copy()
    out = []
    for i in range(n):
        if i in jump_at:
            st = st + np.array([9.0, -7.5, 1.0], np.float32)          # a reset: far outside the window staged around the previous state
        out.append(st.copy())
        st = st + np.array([0.08, 0.06, 0.02], np.float32) * rng.uniform(0.5, 1.5, 3).astype(np.float32)
    return out


def _run(pl, states, T, events=None, sleep_at=(), sleep_s=0.0):
    """forward_state_async + first_action per step, the caller's output block kept per step; `events[i]` runs after step i."""
    import torch
    n_out = T * 2 + (T + 1) * 3
    outs, acts = [], []
    for i, st in enumerate(states):
        if i in sleep_at:
            time.sleep(sleep_s)
        o = torch.full((n_out,), float("nan"), device="cuda")
        torch.cuda.current_stream().synchronize()            # the mode's contract: nothing queued on the planner's stream still uses the block handed over
        pl.forward_state_async(st, None, 0, o.data_ptr())
        acts.append(pl.first_action().copy())
        outs.append(o)
        if events and i in events:
            events[i](pl)
    pl.flush()                                   # ends the loop: the launch waiting for a next state is cancelled
    torch.cuda.synchronize()
    pl.sync()
    return [o.cpu().numpy() for o in outs], acts, (pl.states(), pl.costs(), pl.weights(), pl.get_mean())


@pytest.mark.parametrize("K,T,ref_order,store_u,extra", [(1024, 50, False, False, {}), (1024, 50, True, False, {}), (512, 33, False, True, {}), (128, 20, False, False, {}),
                                                         (100, 7, True, True, {}), (1024, 50, False, False, {"lean": True}), (512, 40, False, False, {"resolution": 0.3}),
                                                         (512, 40, True, True, {"resolution": 0.3})],
                         ids=["c2", "c2-ref", "K512-U", "c1", "ragged-ref-U", "lean", "res0.3", "res0.3-ref-U"])
def test_host_paced_loop_is_bit_identical_to_the_one_launch_loop(K, T, ref_order, store_u, extra):
    from benchnav_amd import synth
    G = 256
    inst = synth.make_instance(G, seed=3, resolution=extra.get("resolution", 0.5))
    states = _states(inst, 14, jump_at=(6,))
    res = {}
    for paced in (True, False):
        with _planner(K, T, G, inst, paced, reference_order=ref_order, store_controls=store_u, **extra) as pl:
            assert pl.host_paced() == paced
            res[paced] = _run(pl, states, T)
            assert pl.solve_count() == len(states)
            if store_u:
                res[paced] += (pl.controls(),)
    for i, (a, b) in enumerate(zip(res[True][0], res[False][0])):
        assert np.array_equal(a, b), ("output block", i)
        assert np.isfinite(a).all()
    for i, (a, b) in enumerate(zip(res[True][1], res[False][1])):
        assert np.array_equal(a, b), ("first action", i)
        assert np.array_equal(a, res[True][0][i][:2])                  # the mailbox IS U*[0]
    for j, (a, b) in enumerate(zip(res[True][2], res[False][2])):
        assert np.array_equal(a, b), ("final", j)
    if store_u:
        assert np.array_equal(res[True][3], res[False][3])


def test_every_other_entry_point_cancels_the_waiting_launch_and_the_loop_goes_on():
    """Getters, setters, a batch of solves and a re-roll in the middle of a host-paced loop: each cancels the launch that waits for the next
    state (the books go back), does its own work, and the loop resumes with an ordinary launch -- same results as without pacing."""
    import torch
    from benchnav_amd import synth
    K, T, G = 1024, 50, 256
    inst = synth.make_instance(G, seed=5)
    states = _states(inst, 12)
    st_dev = torch.from_numpy(states[0]).cuda()
    seen = {True: [], False: []}

    def events_for(paced):
        rec = seen[paced]
        return {1: lambda pl: rec.append(pl.weights()), 3: lambda pl: rec.append(pl.top_samples(5)), 5: lambda pl: pl.solve_n_async_device(4, st_dev.data_ptr()),
                7: lambda pl: pl.set_goal(inst.goal.numpy() - 1.0), 8: lambda pl: rec.append(pl.get_mean()), 9: lambda pl: pl.set_mean(np.full((T, 2), 0.25, np.float32)),
                10: lambda pl: rec.append(pl.solve_count())}
    res = {}
    for paced in (True, False):
        with _planner(K, T, G, inst, paced) as pl:
            res[paced] = _run(pl, states, T, events_for(paced))
    for i, (a, b) in enumerate(zip(res[True][0], res[False][0])):
        assert np.array_equal(a, b), ("output block", i)
    for a, b in zip(res[True][1], res[False][1]):
        assert np.array_equal(a, b)
    for j, (a, b) in enumerate(zip(res[True][2], res[False][2])):
        assert np.array_equal(a, b), ("final", j)
    assert len(seen[True]) == len(seen[False]) == 4
    for a, b in zip(seen[True], seen[False]):
        if isinstance(a, tuple):
            assert all(np.array_equal(x, y) for x, y in zip(a, b))
        else:
            assert np.array_equal(a, b)


@pytest.mark.parametrize("unchecked", [False, True], ids=["seen-before-the-post", "repaired-in-first_action"])
def test_a_launch_that_gave_up_waiting_is_replaced(unchecked):
    """The host takes longer than the waiting launch waits (here: one look, so it is gone before any host can answer).  Either the next
    forward sees that before it posts and starts over with an ordinary launch, or -- the race, forced here by the test hook -- it posts to
    a launch that has just left, and bn_mppi_first_action notices, takes the solve out of the books and runs it again.  Same results."""
    from benchnav_amd import _capi, synth
    K, T, G = 1024, 50, 256
    inst = synth.make_instance(G, seed=6)
    states = _states(inst, 8)
    res = {}
    for paced in (True, False):
        with _planner(K, T, G, inst, paced) as pl:
            if paced:
                _capi.check(pl._lib.bn_mppi_debug_host_paced(pl._h, 1, 1 if unchecked else 0))
            res[paced] = _run(pl, states, T, sleep_at=(2, 3, 5), sleep_s=0.02)
    for i, (a, b) in enumerate(zip(res[True][0], res[False][0])):
        assert np.array_equal(a, b), ("output block", i)
    for a, b in zip(res[True][1], res[False][1]):
        assert np.array_equal(a, b)
    for j, (a, b) in enumerate(zip(res[True][2], res[False][2])):
        assert np.array_equal(a, b), ("final", j)


def test_drop_in_class_host_loop():
    """MPPI(host_loop=True, noise="philox") driven like the reference's loop: forward(cpu_state), first_action(), the outputs used on
    torch's stream while the next launch already waits (they are ordered behind the solve's own launch), get_top_samples in between."""
    import torch
    from helpers import load_case, mppi_for_fixture
    fx = load_case("c2")
    outs = {}
    for loop in (True, "actions", False):
        solver = mppi_for_fixture(fx, noise="philox", store_controls=True, host_loop=loop)
        assert solver._host_loop == bool(loop) and solver._unordered == (loop == "actions")
        state = torch.tensor(fx["state_0"])
        seq = []
        for i in range(9):
            U, X = solver(state)
            a = solver.first_action().clone()
            if i % 2:
                solver.order_outputs()                   # "actions": what forward() returned is ordered from here (a no-op otherwise) ...
            else:
                solver._weights                          # ... and so it is behind any of the planner's own attributes
            seq.append((U.clone(), X.clone(), a, solver._weights.clone(), solver._state_seq_batch[::37].clone(), solver._perturbed_action_seqs[::41].clone()))
            if i == 4:
                seq.append(tuple(t.clone() for t in solver.get_top_samples(7)))
            state = state + torch.tensor([0.07, 0.05, 0.02])
        solver.release()
        torch.cuda.synchronize()
        outs[loop] = [tuple(t.cpu().numpy() for t in rec) for rec in seq]
        assert np.array_equal(outs[loop][0][2], outs[loop][0][0][0])
    for loop in (True, "actions"):
        for i, (ra, rb) in enumerate(zip(outs[loop], outs[False])):
            for j, (a, b) in enumerate(zip(ra, rb)):
                assert np.array_equal(a, b), (loop, i, j)


def test_unordered_outputs_are_ordered_by_the_first_call_that_needs_them():
    """BN_FLAG_UNORDERED_OUTPUTS: the steady-state forward leaves the handle's stream alone; bn_mppi_order_outputs, or any entry point that
    enqueues on the stream or hands out results, orders it behind the latest posted solve.  Copies taken on the stream right behind such a
    call, in the middle of the loop, hold the finished block; the loop's results are those of the one-launch loop."""
    import torch
    from benchnav_amd import synth
    K, T, G = 1024, 50, 256
    inst = synth.make_instance(G, seed=11)
    states = _states(inst, 16, jump_at=(9,))
    n_out = T * 2 + (T + 1) * 3
    res = {}
    for paced in (True, False):
        with _planner(K, T, G, inst, paced, unordered_outputs=paced) as pl:
            assert pl.host_paced() == paced
            copies, acts, blocks = [], [], []
            for i, st in enumerate(states):
                o = torch.full((n_out,), float("nan"), device="cuda")
                torch.cuda.current_stream().synchronize()
                pl.forward_state_async(st, None, 0, o.data_ptr())
                acts.append(pl.first_action().copy())
                blocks.append(o)
                if i % 3 == 0:
                    pl.order_outputs()
                    copies.append(o.clone())                 # on the handle's stream (stream=0), right behind the ordering call
                elif i % 3 == 1:
                    w = pl.weights()                         # a getter: orders (and cancels the waiting launch)
                    copies.append(o.clone())
                    copies.append(torch.from_numpy(w))
            pl.flush(); torch.cuda.synchronize(); pl.sync()
            res[paced] = ([c.cpu().numpy() for c in copies], acts, [b.cpu().numpy() for b in blocks], (pl.states(), pl.costs(), pl.weights(), pl.get_mean()))
    for k in range(4):
        assert len(res[True][k]) == len(res[False][k])
        for i, (a, b) in enumerate(zip(res[True][k], res[False][k])):
            assert np.array_equal(a, b), (k, i)
    assert all(np.isfinite(c).all() for c in res[True][0])


def test_forwards_fired_back_to_back_are_held_to_the_devices_pace():
    """No first_action between the forwards: the host runs ahead of the device, and the request words of a slot must not be written again
    before the launch four tags back has taken its own (the tail workgroups acknowledge; the host waits).  60 forwards, results as ever."""
    import torch
    from benchnav_amd import synth
    K, T, G = 1024, 50, 256
    inst = synth.make_instance(G, seed=7)
    states = _states(inst, 60)
    res = {}
    for paced in (True, False):
        with _planner(K, T, G, inst, paced) as pl:
            for st in states:
                pl.forward_state_async(st)
            fa = pl.first_action().copy()
            pl.flush(); torch.cuda.synchronize(); pl.sync()
            res[paced] = (fa, pl.states(), pl.costs(), pl.weights(), pl.get_mean())
    for j, (a, b) in enumerate(zip(res[True], res[False])):
        assert np.array_equal(a, b), j


def test_handles_that_do_not_qualify_say_so():
    from benchnav_amd import NativeMPPI
    for kw in (dict(num_samples=2048), dict(num_samples=512, num_instances=2), dict(num_samples=512, overlap=False), dict(num_samples=512, sampled_slip=True)):
        args = dict(horizon=20, grid_size=64, resolution=0.5, host_paced=True)
        args.update(kw)
        with NativeMPPI(**args) as pl:
            assert not pl.host_paced()
