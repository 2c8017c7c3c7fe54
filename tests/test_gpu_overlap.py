"""GPU: overlapped launches of bn_mppi_solve_n_async (consecutive solves alternate between two streams; device counters
carry the dependency; per-solve buffers rotate over three slots).  Bar: bit-identical to the one-stream chain -- every output
of the last solve, and through it (warm start) every solve before it -- for every noise source, lean mode, several
instances, and call sequences that mix overlapped batches with single solves and getters."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = 256


def _make(K, T, B, insts, overlap, **kw):
    from benchnav_amd import NativeMPPI
    pl = NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, num_instances=B, seed=5, overlap=overlap, **kw)
    for b, it in enumerate(insts):
        pl.set_map(it.risk.numpy(), b); pl.set_goal(it.goal.numpy(), b)
    return pl


def _outputs(pl, B, T):
    import torch
    from benchnav_amd import _capi
    from benchnav_amd.mppi import _DevArray
    pl.sync()
    xs = torch.as_tensor(_DevArray(pl.device_buffer(_capi.BN_BUF_XSTAR)[0], (B, T + 1, 3)), device="cuda").cpu().numpy()
    us = torch.as_tensor(_DevArray(pl.device_buffer(_capi.BN_BUF_USTAR)[0], (B, T, 2)), device="cuda").cpu().numpy()
    return [(pl.states(b), pl.costs(b), pl.weights(b), pl.get_mean(b), xs[b].copy(), us[b].copy())
            + ((pl.controls(b),) if pl.store_controls else ()) for b in range(B)]


@pytest.mark.parametrize("K,T,B,noise,lean,n", [(1024, 50, 1, "philox", False, 40), (1024, 50, 3, "philox", True, 25), (1000, 33, 2, "kt2", False, 12),
                                                (2048, 50, 1, "t2k", False, 9), (130, 7, 5, "philox", False, 60), (64, 1, 1, "philox", False, 30),
                                                (1024, 50, 1, "philox", False, 3),
                                                # the ticket path (K > 4096: merge by the last workgroup, counted in for the overlapped successor)
                                                (5000, 50, 1, "philox", False, 12), (16384, 100, 1, "philox", False, 7), (8192, 33, 2, "kt2", True, 9),
                                                (4160, 20, 3, "t2k", False, 16),
                                                # the edges of the pre-drawn noise (fast prologue, Philox: step pairs behind chunks 0 and 1 are drawn into the
                                                # tile before the predecessor's rows arrive): no pair, half a pair, ragged chunks, the last horizon of the fast
                                                # prologue (2T = 128) and the first one behind it
                                                (256, 8, 1, "philox", False, 12), (256, 9, 2, "philox", False, 12), (320, 11, 1, "philox", False, 12),
                                                (256, 13, 1, "philox", True, 12), (256, 63, 1, "philox", False, 8), (256, 64, 1, "philox", False, 8),
                                                (256, 65, 1, "philox", False, 8)],
                         ids=["c2", "c2-B3-lean", "ragged-kt2", "K2048-t2k", "small-B5", "T1", "n3", "ticket-ref5000", "ticket-c5", "ticket-B2-lean", "ticket-B3-n16",
                              "predraw-T8", "predraw-T9-B2", "predraw-T11", "predraw-T13-lean", "predraw-T63", "predraw-T64", "predraw-T65"])
def test_overlapped_chain_equals_one_stream_chain(K, T, B, noise, lean, n):
    import torch
    from benchnav_amd import _capi, synth
    insts = [synth.make_instance(G, seed=70 + b, jitter=True) for b in range(B)]
    st = torch.stack([it.start for it in insts]).cuda()
    ring = min(n, 4)
    eps = np.random.default_rng(6).standard_normal((ring, B, K, T, 2)).astype(np.float32)
    if noise == "kt2":
        ed, kind = torch.from_numpy(eps).cuda(), _capi.BN_NOISE_DEVICE_KT2
    elif noise == "t2k":
        ed, kind = torch.from_numpy(np.ascontiguousarray(eps.transpose(0, 1, 3, 4, 2))).cuda(), _capi.BN_NOISE_DEVICE_T2K
    else:
        ed, kind = None, _capi.BN_NOISE_PHILOX
    torch.cuda.synchronize()
    res = {}
    for overlap in (False, True):
        with _make(K, T, B, insts, overlap, store_controls=not lean, lean=lean, kernel="lat") as pl:
            if ed is None:
                pl.solve_n_async_device(n, st.data_ptr())
            else:
                pl.solve_n_async_device(n, st.data_ptr(), ed.data_ptr(), kind, ring, eps[0].size)
            res[overlap] = _outputs(pl, B, T)
    for b in range(B):
        for j, (a_, b_) in enumerate(zip(res[True][b], res[False][b])):
            assert np.array_equal(a_, b_), (b, j)


@pytest.mark.parametrize("kernel,K,T,B,noise,lean,n", [("role", 1024, 50, 2, "philox", False, 9), ("role", 1000, 33, 3, "kt2", False, 6),
                                                        ("role", 512, 20, 40, "philox", False, 8), ("role", 1024, 50, 70, "philox", True, 7),
                                                        ("role", 256, 12, 300, "philox", False, 5),
                                                        ("wave", 1024, 50, 3, "philox", False, 8), ("wave", 1000, 33, 2, "t2k", False, 6),
                                                        ("wave", 512, 20, 150, "philox", False, 6), ("wave", 512, 20, 150, "philox", True, 5)],
                         ids=["role-B2", "role-ragged-kt2", "role-B40", "role-B70-lean", "role-B300-two-rounds", "wave-B3", "wave-ragged-t2k", "wave-B150",
                              "wave-B150-lean"])
def test_many_instance_kernels_overlap_their_launches_bit_identically(kernel, K, T, B, noise, lean, n):
    """The role kernel and the one-wave throughput kernel publish per instance and wait per instance: a workgroup of solve i+1
    takes the slot a finished workgroup of solve i leaves and waits there for its own instance only.  Same bar as above."""
    import torch
    from benchnav_amd import _capi, synth
    base = [synth.make_instance(G, seed=90 + b, jitter=True) for b in range(min(B, 6))]
    insts = [base[b % len(base)] for b in range(B)]
    st = torch.stack([it.start for it in insts]).clone()
    st[:, 0] += torch.arange(B) * 0.01                      # instances differ even where they share a map
    st = st.cuda()
    ring = min(n, 3)
    ed, kind, stride = None, _capi.BN_NOISE_PHILOX, 0
    if noise != "philox":
        eps = np.random.default_rng(8).standard_normal((ring, B, K, T, 2)).astype(np.float32)
        stride = eps[0].size
        if noise == "kt2":
            ed, kind = torch.from_numpy(eps).cuda(), _capi.BN_NOISE_DEVICE_KT2
        else:
            ed, kind = torch.from_numpy(np.ascontiguousarray(eps.transpose(0, 1, 3, 4, 2))).cuda(), _capi.BN_NOISE_DEVICE_T2K
    torch.cuda.synchronize()
    res = {}
    for overlap in (False, True):
        with _make(K, T, B, insts, overlap, store_controls=not lean, lean=lean, kernel=kernel) as pl:
            if ed is None:
                pl.solve_n_async_device(n, st.data_ptr())
            else:
                pl.solve_n_async_device(n, st.data_ptr(), ed.data_ptr(), kind, ring, stride)
            pick = sorted(set([0, B // 3, B // 2, B - 1]))
            out = _outputs(pl, B, T)
            res[overlap] = [out[b] for b in pick]
    for a, c in zip(res[True], res[False]):
        for j, (a_, c_) in enumerate(zip(a, c)):
            assert np.array_equal(a_, c_), j


@pytest.mark.parametrize("K,T,B,n,corner", [(1024, 50, 1, 40, False), (512, 20, 3, 60, True), (256, 12, 40, 30, False), (1024, 50, 20, 12, False)],
                         ids=["c2", "B3-map-corner", "B40", "role-B20"])
def test_overlapped_episodes_log_the_same_closed_loop(K, T, B, n, corner):
    """Device-side episodes with overlapped launches: the successor reads the state its predecessor advanced (counter path:
    device-scope loads after the wait; latency kernel: three tagged granules, polled EARLY so that the latent-slip loads and a
    window one environment step wider are in flight before the control is known).  Same log, same final solve as one stream,
    for the latency kernel, the role kernel, and starts in a map corner (the wider window is shifted into the map)."""
    import torch
    from benchnav_amd import synth
    base = [synth.make_instance(G, seed=40 + b, jitter=True) for b in range(min(B, 4))]
    insts = [base[b % len(base)] for b in range(B)]
    starts = np.stack([it.start.numpy() for it in insts]).astype(np.float32)
    starts[:, 0] += np.arange(B, dtype=np.float32) * 0.02
    if corner:
        starts[0, :2] = [0.2, 0.3]; starts[-1, :2] = [G * 0.5 - 0.3, G * 0.5 - 0.2]
    lat = np.stack([it.risk.numpy() for it in insts]); std = np.full_like(lat, 0.08)
    res = {}
    for overlap in (False, True):
        with _make(K, T, B, insts, overlap) as pl:
            pl.env_attach(lat, std, goal_threshold=1.0, delta_t=0.1, seed=77)
            log = pl.episode(n, starts)
            res[overlap] = (log, pl.last_actions.copy(), _outputs(pl, B, T))
    (s0, r0, d0), a0, o0 = res[False]
    (s1, r1, d1), a1, o1 = res[True]
    assert np.array_equal(s0, s1) and np.array_equal(r0, r1) and np.array_equal(d0, d1) and np.array_equal(a0, a1)
    assert np.abs(np.diff(s0[:, :, :2], axis=0)).max() > 1e-3                     # the rovers do move
    for b in (0, B - 1):
        for j, (x, y) in enumerate(zip(o1[b], o0[b])):
            assert np.array_equal(x, y), (b, j)


def test_speculative_window_on_a_map_it_fills_exactly():
    """Closed loop, latency kernel, overlapped: the successor stages a window one environment step wider around the PREVIOUS state
    (spec_extra cells each side).  On a 24 x 24 map that window is the whole map plus the guard row / column (WN + 2 spec_extra =
    25 = G + 1, the largest bn_mppi_env_attach admits): rovers in both corners, same log as on one stream."""
    from benchnav_amd import NativeMPPI, synth
    K, T, B, n, Gs = 1024, 50, 2, 30, 24
    inst = synth.make_instance(Gs, seed=3)
    starts = np.array([[0.3, 0.4, 0.5], [Gs * 0.5 - 0.25, Gs * 0.5 - 0.35, 0.6]], np.float32)     # lower-left and upper-right corner
    lat = inst.risk.numpy() * 0.5; std = np.full_like(lat, 0.05)
    logs = {}
    for overlap in (False, True):
        with NativeMPPI(horizon=T, num_samples=K, grid_size=Gs, resolution=0.5, num_instances=B, shared_map=True, seed=9, overlap=overlap) as pl:
            pl.set_map(inst.risk.numpy()); pl.set_goal(np.array([Gs * 0.25, Gs * 0.25], np.float32))
            pl.env_attach(lat, std, goal_threshold=0.5, delta_t=0.1, seed=5)
            logs[overlap] = pl.episode(n, starts) + (pl.last_actions.copy(),)
    for a, b_ in zip(logs[False], logs[True]):
        assert np.array_equal(a, b_)
    assert logs[True][0][:, 1, :2].max() >= Gs * 0.5 - 0.4          # the second rover stays near the upper limits for a while


def test_big_batches_behind_pending_work_do_not_crowd_each_other_out():
    """A waiting workgroup holds its slot, and two launches that become eligible at the same instant are not dispatched
    fairly: the successor's waiting workgroups can take every slot before its predecessor got one.  Inside a batch that cannot
    happen (a launch becomes eligible a whole kernel after its predecessor); at the start of a batch it could when earlier
    work is still pending on the stream -- big launches wait for the stream there.  70 instances (more workgroups than slots)
    behind a few milliseconds of matmuls, and two batches back to back: no wait expires, results as on one stream."""
    import torch
    from benchnav_amd import NativeMPPI, synth
    K, T, B = 1024, 50, 70
    inst = synth.make_instance(G, seed=21)
    st = torch.stack([inst.start + torch.tensor([0.01 * b, 0.0, 0.0]) for b in range(B)]).cuda()
    stream = torch.cuda.Stream()
    big = torch.randn(2048, 2048, device="cuda")
    torch.cuda.synchronize()
    res = {}
    for overlap in (False, True):
        with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, num_instances=B, shared_map=True, seed=5, lean=True,
                        overlap=overlap, kernel="role", stream=stream.cuda_stream) as pl:
            pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
            with torch.cuda.stream(stream):
                acc = big
                for _ in range(30):
                    acc = (acc @ big) * 1e-3                # keeps the stream busy while the batches are enqueued
            pl.solve_n_async_device(6, st.data_ptr())
            pl.solve_n_async_device(7, st.data_ptr())        # behind the first batch, no sync in between
            pl.sync()                                        # raises if a device-side wait gave up
            res[overlap] = [(pl.costs(b), pl.weights(b), pl.get_mean(b)) for b in (0, B // 2, B - 1)]
            assert pl.solve_count() == 13
    for a, c in zip(res[True], res[False]):
        for j, (a_, c_) in enumerate(zip(a, c)):
            assert np.array_equal(a_, c_), j


def test_an_expired_wait_is_repaired_by_a_rerun_on_one_stream():
    """A bounded device-side wait that expires (another process keeping a predecessor off the chip) spoils the batches enqueued
    since the last synchronisation point.  Every synchronising entry point -- a getter here, not only sync() -- notices the error
    word and re-runs those batches on one stream from the mean the first of them started from.  With the test hook (which also
    overwrites everything the batches wrote with NaN patterns): two chained batches behind a warm-up, then weights / costs / mean /
    trajectories equal to a handle that never overlapped; the event is counted, warned about once, and the handle keeps to one
    stream (and keeps working) afterwards."""
    import torch
    from benchnav_amd import _capi, synth
    K, T, B, n = 1024, 50, 2, 6
    insts = [synth.make_instance(G, seed=60 + b) for b in range(B)]
    st = torch.stack([it.start for it in insts]).cuda()
    eps = np.random.default_rng(12).standard_normal((n, B, K, T, 2)).astype(np.float32)
    ed = torch.from_numpy(eps).cuda()
    torch.cuda.synchronize()
    run = lambda pl, m=n: pl.solve_n_async_device(m, st.data_ptr(), ed.data_ptr(), _capi.BN_NOISE_DEVICE_KT2, n, eps[0].size)
    with _make(K, T, B, insts, False) as ref:
        run(ref); run(ref, 4); run(ref, 5)
        want = _outputs(ref, B, T)
        run(ref, 3)
        want2 = _outputs(ref, B, T)
    with _make(K, T, B, insts, True) as pl:
        run(pl)                                              # a batch that is checked clean ...
        pl.sync()
        assert pl.recovery_count() == 0
        run(pl, 4); run(pl, 5)                               # ... and two, chained, that are not
        _capi.check(pl._lib.bn_mppi_debug_expire_wait(pl._h))
        w0 = pl.weights(0)                                   # a getter, not sync(): repaired here
        assert np.isfinite(w0).all()
        assert pl.recovery_count() == 1
        assert b"re-run on one stream" in pl._lib.bn_last_error()
        got = _outputs(pl, B, T)
        assert pl.recovery_count() == 1                      # repaired once
        run(pl, 3)                                           # one stream now; the chain goes on from the repaired mean
        got2 = _outputs(pl, B, T)
        assert pl.recovery_count() == 1
    for g_, w_ in ((got, want), (got2, want2)):
        for b in range(B):
            for j, (a_, c_) in enumerate(zip(g_[b], w_[b])):
                assert np.array_equal(a_, c_), (b, j)


@pytest.mark.parametrize("K,kw", [(1024, {}), (1024, {"kernel": "role"}), (1024, {"kernel": "wave"}), (8192, {}),
                                  (1024, {"reference_order": True})],
                         ids=["lat", "role", "wave", "ticket", "reference-order"])
def test_a_rerun_does_not_read_the_callers_state_buffer_again(K, kw):
    """A host-driven closed loop reuses ONE device state buffer: batch, write the next state into it in stream order, batch.  Both
    batches have read what they needed by the time a later synchronisation point notices an expired wait -- the re-run must start
    each of them from the states it was GIVEN, not from what the buffer holds by then.  The first launch of every journalled batch
    keeps its states in handle-owned memory (SolveParams::state_snap); with the test hook: three chained batches from three
    different states in the same buffer, then the buffer is scribbled over, then the repair -- results equal to a handle that never
    overlapped.  The same for a single solve behind the batches; on every kernel family, since each keeps the snapshot itself."""
    import torch
    from benchnav_amd import _capi, synth
    T, B = 50, 2
    insts = [synth.make_instance(G, seed=64 + b) for b in range(B)]
    st0 = torch.stack([it.start for it in insts])
    states = [st0, st0 + torch.tensor([0.4, -0.3, 0.2]), st0 + torch.tensor([-0.5, 0.6, -0.4]), st0 + torch.tensor([0.1, 0.2, 0.7])]
    buf = torch.empty_like(st0).cuda()
    torch.cuda.synchronize()

    def drive(pl):
        for i, n in enumerate((4, 5, 3)):
            torch.cuda.synchronize()                             # the device, not the handle: its journal stays unchecked
            buf.copy_(states[i]); torch.cuda.synchronize()
            pl.solve_n_async_device(n, buf.data_ptr())
        torch.cuda.synchronize()
        buf.copy_(states[3]); torch.cuda.synchronize()
        pl.solve_async_device(buf.data_ptr())                    # a single solve behind them
        torch.cuda.synchronize()

    with _make(K, T, B, insts, False, **kw) as ref:
        drive(ref)
        want = _outputs(ref, B, T)
    with _make(K, T, B, insts, True, **kw) as pl:
        drive(pl)
        _capi.check(pl._lib.bn_mppi_debug_expire_wait(pl._h))    # (synchronises the handle's streams first)
        buf.fill_(float("nan"))                                  # the caller moves on: nothing the batches were given is left
        torch.cuda.synchronize()
        got = _outputs(pl, B, T)
        assert pl.recovery_count() == 1
    for b in range(B):
        for j, (a_, c_) in enumerate(zip(got[b], want[b])):
            assert np.array_equal(a_, c_), (b, j)


def test_an_expired_wait_in_an_episode_is_repaired():
    """The device-side closed loop is journalled as a whole: after the hook the log read-back equals the one-stream episode's."""
    from benchnav_amd import _capi, synth
    K, T, B, steps = 512, 20, 2, 12
    insts = [synth.make_instance(G, seed=70 + b) for b in range(B)]
    logs = {}
    for overlap in (False, True):
        with _make(K, T, B, insts, overlap) as pl:
            pl.env_attach(np.stack([it.risk.numpy() for it in insts]), np.full((B, G, G), 0.05, np.float32))
            st0 = np.stack([it.start.numpy() for it in insts])
            pl.episode(steps, st0, wait=False)
            if overlap:
                _capi.check(pl._lib.bn_mppi_debug_expire_wait(pl._h))
            logs[overlap] = pl.episode_log()
            if overlap:
                assert pl.recovery_count() == 1
    for a_, c_ in zip(logs[True], logs[False]):
        assert np.array_equal(a_, c_)


def test_two_handles_in_flight_do_not_starve_each_other():
    """Workgroups of an overlapped launch hold their slots while they wait.  Several handles doing that at once can leave no
    slot for each other's predecessors; the library lets one handle per device overlap at a time and runs the other's batch
    in one stream.  Eight planners enqueue long batches back to back without a sync in between: no wait expires (sync() would
    raise), and each ends where its own one-stream run ends."""
    import torch
    from benchnav_amd import synth
    K, T, B, n = 1024, 50, 8, 60
    insts = [synth.make_instance(G, seed=11 + b) for b in range(B)]
    st = torch.stack([it.start for it in insts]).cuda()
    torch.cuda.synchronize()
    with _make(K, T, B, insts, False) as ref:
        ref.solve_n_async_device(n, st.data_ptr())
        want = _outputs(ref, B, T)
    planners = [_make(K, T, B, insts, True) for _ in range(8)]
    try:
        for rep in range(2):
            for pl in planners:
                pl.solve_n_async_device(n // 2, st.data_ptr())
        for pl in planners:
            got = _outputs(pl, B, T)                     # sync() inside raises if a device-side wait gave up
            for b in (0, B - 1):
                for j, (a_, c_) in enumerate(zip(got[b], want[b])):
                    assert np.array_equal(a_, c_), (b, j)
    finally:
        for pl in planners:
            pl.close()


def test_fresh_handles_keep_the_last_solves_trajectories():
    """Two launches in flight never write the same trajectory / control addresses (the batch alternates between two buffers and
    ends on the exposed one).  A cold handle is where a late-ending earlier kernel used to overwrite rows of the last solve:
    repeat short batches on fresh handles, odd and even lengths, and compare the dumps with the one-stream chain."""
    import torch
    from benchnav_amd import _capi, synth
    K, T, B = 1000, 33, 2
    insts = [synth.make_instance(G, seed=30 + b, jitter=True) for b in range(B)]
    st = torch.stack([it.start for it in insts]).cuda()
    eps = np.random.default_rng(4).standard_normal((6, B, K, T, 2)).astype(np.float32)
    ed = torch.from_numpy(eps).cuda()
    torch.cuda.synchronize()
    for n in (5, 6):
        want = None
        for rep in range(7):
            with _make(K, T, B, insts, rep > 0, store_controls=True, kernel="lat") as pl:
                pl.solve_n_async_device(n, st.data_ptr(), ed.data_ptr(), _capi.BN_NOISE_DEVICE_KT2, n, eps[0].size)
                got = _outputs(pl, B, T)
            if want is None:
                want = got
                continue
            for b in range(B):
                for j, (a_, b_) in enumerate(zip(got[b], want[b])):
                    assert np.array_equal(a_, b_), (n, rep, b, j)


@pytest.mark.parametrize("reference_order", [False, True], ids=["spec", "reference-order"])
def test_mixed_call_sequences_and_a_long_chain(reference_order):
    """Batches of different lengths, single solves, getters and a changed state in between; then 3000 solves in one call.
    (Round 5: the tails of overlapped solves order only their stores -- default arithmetic -- or stage their window before their rows
    are there and check the state behind the wait -- reference order: the batches that change the start state are what a stale
    window would show up in, through X*.)"""
    import torch
    from benchnav_amd import synth
    K, T = 512, 20
    inst = synth.make_instance(G, seed=3)
    st = inst.start.cuda()
    st2 = (inst.start + torch.tensor([0.3, -0.2, 0.1])).cuda()
    torch.cuda.synchronize()
    res = {}
    for overlap in (False, True):
        with _make(K, T, 1, [inst], overlap, kernel="lat", reference_order=reference_order) as pl:
            seq = []
            pl.solve_n_async_device(7, st.data_ptr())
            seq.append(pl.get_mean(0))                       # getter: flushes the pending tail
            pl.solve_async_device(st2.data_ptr())            # a single solve between batches (one stream)
            pl.solve_n_async_device(4, st2.data_ptr())
            seq.append(pl.weights(0))
            seq.extend(_outputs(pl, 1, T)[0])                # (X* of a batch whose slots still hold the other state)
            pl.solve_n_async_device(40, st.data_ptr())
            pl.solve_n_async_device(25, st2.data_ptr())      # two long batches back to back, the state changes between them
            seq.extend(_outputs(pl, 1, T)[0])
            pl.solve_n_async_device(3, st.data_ptr())
            pl.solve_n_async_device(2, st.data_ptr())        # too short to overlap
            seq.append(pl.costs(0))
            pl.solve_n_async_device(3000, st.data_ptr())
            seq.extend(_outputs(pl, 1, T)[0])
            assert pl.solve_count() == 7 + 1 + 4 + 40 + 25 + 3 + 2 + 3000
            res[overlap] = seq
    for j, (a_, b_) in enumerate(zip(res[True], res[False])):
        assert np.array_equal(a_, b_), j


def test_sampled_slip_launches_overlap_bit_identically():
    """BASELINE config 3's kernel on the ticket path: overlapped launches (slip draws before the wait for the predecessor's merged
    mean, controls behind it) give the chain of the one-stream launches, bit for bit."""
    import torch
    from benchnav_amd import NativeMPPI, synth
    K, T = 8192, 50
    inst = synth.make_instance(G, seed=5)
    st = inst.start.cuda()
    res = {}
    for overlap in (False, True):
        with NativeMPPI(horizon=T, num_samples=K, grid_size=G, resolution=0.5, sampled_slip=True, seed=9, overlap=overlap) as pl:
            pl.set_map(inst.risk.numpy()); pl.set_slip_std(synth.slip_std_map(G, seed=5).numpy()); pl.set_goal(inst.goal.numpy())
            pl.solve_n_async_device(9, st.data_ptr())
            pl.sync()
            res[overlap] = (pl.states(), pl.costs(), pl.weights(), pl.get_mean())
            assert pl.solve_count() == 9
    for a_, b_ in zip(res[True], res[False]):
        assert np.array_equal(a_, b_)


def test_overlap_mode_getter_and_the_tuners_policy():
    """bn_mppi_overlap_mode (ABI 4) and the policy behind it, fed through the test hook: a handle looks at the one-stream mode once, stays
    overlapped while that is faster, moves to one stream when its own cadence degrades by half AND one stream beats it (a co-tenant on
    the device), looks back every 256 windows, and returns when the device is its own again.  An expired wait ends overlapping for good."""
    import torch
    from benchnav_amd import NativeMPPI, _capi, synth
    inst = synth.make_instance(128, seed=1)
    st = inst.start.cuda()
    torch.cuda.synchronize()
    with NativeMPPI(horizon=30, num_samples=512, grid_size=128, resolution=0.5) as pl:
        pl.set_map(inst.risk.numpy()); pl.set_goal(inst.goal.numpy())
        feed = lambda mode, us: pl._lib.bn_mppi_debug_cadence(pl._h, mode, us)
        assert pl.overlap_mode() == 0
        for i in range(63):
            assert feed(0, 8.0 + 0.001 * i) == 0
        assert feed(0, 8.1) == 1                                 # 64 windows seen (4096 launches): the next long batch looks at one stream
        assert feed(1, 12.3) == 0 and pl.overlap_mode() == 0     # slower: stay
        assert feed(0, 9.0) == 0 and pl.overlap_mode() == 0
        assert feed(0, 3.0) == 0 and feed(0, 8.0) == 0 and pl.overlap_mode() == 0   # one implausibly short window does not become the yardstick
        for i in range(3):
            feed(0, 21.0)                                        # 2.6 x the best overlapped cadence and well above one stream's ...
            assert pl.overlap_mode() == 0
        feed(0, 21.0)                                            # ... four windows in a row: a co-tenant
        assert pl.overlap_mode() == 1
        for i in range(255):
            assert feed(1, 12.5) == 0 and pl.overlap_mode() == 1
        assert feed(1, 12.5) == 1                                # 256 windows on one stream: look at the overlapped mode again
        feed(0, 20.0)
        assert pl.overlap_mode() == 1                            # still there
        for i in range(256):
            feed(1, 12.5)
        feed(0, 8.2)                                             # gone
        assert pl.overlap_mode() == 0
        # batches run and give the same results whatever mode the tuner picks
        pl.solve_n_async_device(200, st.data_ptr()); pl.sync()
        a = pl.get_mean()
        for i in range(4):
            feed(0, 30.0)
        assert pl.overlap_mode() == 1
        pl.set_mean(None)
    with NativeMPPI(horizon=30, num_samples=512, grid_size=128, resolution=0.5) as p2, \
         NativeMPPI(horizon=30, num_samples=512, grid_size=128, resolution=0.5, overlap=False) as p3:
        for q in (p2, p3):
            q.set_map(inst.risk.numpy()); q.set_goal(inst.goal.numpy())
        assert p3.overlap_mode() == 3
        p2._lib.bn_mppi_debug_cadence(p2._h, 0, 8.0); p2._lib.bn_mppi_debug_cadence(p2._h, 1, 9.0)
        for i in range(4):
            p2._lib.bn_mppi_debug_cadence(p2._h, 0, 40.0)
        assert p2.overlap_mode() == 1                            # by choice: its batches take the one-stream path
        p2.solve_n_async_device(200, st.data_ptr()); p2.sync()
        p3.solve_n_async_device(200, st.data_ptr()); p3.sync()
        assert np.array_equal(p2.get_mean(), p3.get_mean()) and np.array_equal(p2.get_mean(), a)
        assert np.array_equal(p2.states(), p3.states())
    with NativeMPPI(horizon=30, num_samples=512, grid_size=128, resolution=0.5) as p4:
        p4.set_map(inst.risk.numpy()); p4.set_goal(inst.goal.numpy())
        p4.solve_n_async_device(8, st.data_ptr())
        _capi.check(p4._lib.bn_mppi_debug_expire_wait(p4._h))
        p4.sync()
        assert p4.recovery_count() == 1 and p4.overlap_mode() == 2
