"""GPU: video.Video.encode_frame (P3) through the C ABI.

Bit-exact against (a) opcode streams / final state / RNG positions recorded from
the imported reference (tests/golden/g3_encode_runs.npz) and (b) the oracle on
fresh seeded inputs, including batches of independent streams."""

import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = None


def _tags(g3):
    return sorted(set(k.split("/")[0] for k in g3.files))


def _seed_states(O, seed_py, seed_np):
    return O.mt_seed_py(seed_py).state_words(), O.mt_seed_np(seed_np).state_words()


def _run_device(native, device_tables, mode, pal, frames_list, sched, seeds, recurrence=True, wave=True,
                prefix_sort=True):
    """frames_list: list (per stream) of (n_frames, banks, 32, 256) arrays."""
    import torch
    t, s = device_tables.get(mode, pal)
    n = len(frames_list)
    enc = native.Encoder(mode, t, s, n, dm=device_tables.dm[(mode, pal)])
    enc.set_diff_weights_mode(recurrence)
    enc.set_greedy_kernel(wave)
    enc.set_prefix_sort(prefix_sort)
    fr = np.stack(frames_list)
    fm = torch.from_numpy(np.ascontiguousarray(fr[:, :, 0])).cuda()
    fa = torch.from_numpy(np.ascontiguousarray(fr[:, :, 1])).cuda() if mode == 1 else None
    enc.set_state_all(native.STATE_RNG_PY, np.stack([py for py, _ in seeds]))
    enc.set_state_all(native.STATE_RNG_NP, np.stack([npw for _, npw in seeds]))
    segs = [(int(f), int(a), 1, int(k)) for (f, a, k) in sched]
    ops = enc.encode(fm, fa, segs)
    enc.check()
    return enc, ops.cpu().numpy()


def _next_draws(O, words, n, high):
    m = O.MT()
    m.set_state_words(words)
    L = O.lib()
    f = L.orc_py_getrandbits8 if high else L.orc_np_randint256
    return [f(C.byref(m)) for _ in range(n)]


@pytest.mark.parametrize("recurrence,wave", [(True, True), (False, True), (True, False), (False, False), (True, "team"),
                                             ("split", True), ("split", "team"), ("split", False), (True, "shared"),
                                             (False, "shared")])
def test_golden_runs(native, O, golden, device_tables, recurrence, wave):
    """recurrence=True: diff weights recomputed in the prologue; False: gathered from
    the HBM table; "split": combined from the two halves of the split diff-weight table.  wave=True: one wave per stream reading the split store table; False: one
    256-thread workgroup per stream reading the dense u16 store table; "team": eight waves per
    stream scoring the next list entries concurrently.  Every combination must reproduce the
    reference bit for bit."""
    g3 = golden.g3_encode_runs
    for tag in _tags(g3):
        mode, pal, sp, sn = (int(x) for x in g3[tag + "/meta"])
        frames, sched, ops = g3[tag + "/frames"], g3[tag + "/schedule"], g3[tag + "/ops"]
        enc, got = _run_device(native, device_tables, mode, pal, [frames], sched, [_seed_states(O, sp, sn)],
                               recurrence=recurrence, wave=wave)
        bad = np.nonzero((got[0] != ops).any(axis=1))[0]
        assert len(bad) == 0, "%s: first mismatch at op %d: got %s want %s" % (
            tag, bad[0], got[0][bad[0]], ops[bad[0]])
        assert (enc.get_state(native.STATE_MEM_MAIN) == g3[tag + "/mem_main"]).all(), tag
        assert (enc.get_state(native.STATE_UP_MAIN) == g3[tag + "/up_main"]).all(), tag
        assert (enc.get_state(native.STATE_PACKED) == g3[tag + "/packed"]).all(), tag
        if mode == 1:
            assert (enc.get_state(native.STATE_MEM_AUX) == g3[tag + "/mem_aux"]).all(), tag
            assert (enc.get_state(native.STATE_UP_AUX) == g3[tag + "/up_aux"]).all(), tag
        assert enc.get_state(native.STATE_OUT_OF_WORK).tolist() == g3[tag + "/out_of_work"].tolist(), tag
        assert _next_draws(O, enc.get_state(native.STATE_RNG_PY), 4, True) == g3[tag + "/py_next"].tolist(), tag
        assert _next_draws(O, enc.get_state(native.STATE_RNG_NP), 4, False) == g3[tag + "/np_next"].tolist(), tag
        enc.close()


def _synth(mode, n_frames, seed, coherent=False):
    holes = (np.arange(256) & 127) >= 120
    rng = np.random.default_rng(seed)
    hi = 128 if mode == 1 else 256
    banks = 2 if mode == 1 else 1
    fr = np.zeros((n_frames, 2, 32, 256), np.uint8)
    prev = None
    for f in range(n_frames):
        cur = []
        for b in range(banks):
            new = rng.integers(0, hi, (32, 256), dtype=np.uint8)
            if coherent and prev is not None:
                new = np.where(rng.random((32, 256)) < 0.9, prev[b], new)
            new[:, holes] = 0
            cur.append(new)
        prev = cur
        for b in range(banks):
            fr[f, b] = cur[b]
    return fr


def _oracle_run(O, oracle_tables, mode, pal, frames, sched, sp, sn):
    v = O.Video(mode, oracle_tables.get(mode, pal), seed_py=sp, seed_np=sn)
    out = []
    for fi, ia, n in sched:
        v.encode_frame(frames[fi, 0], frames[fi, 1] if mode == 1 else None, ia)
        out.append(v.next(int(n)))
    return v, np.concatenate(out)


@pytest.mark.parametrize("mode,wave,prefix", [(1, True, True), (0, True, True), (1, False, True), (1, True, False), (1, "shared", True), (1, "shared", False), (0, "shared", True),
                                              (0, False, False), (1, "team", True), (0, "team", False)])
def test_batch_of_independent_streams(native, O, oracle_tables, device_tables, mode, wave, prefix):
    """12 streams with different data / seeds / coherence in ONE launch sequence,
    ragged segment lengths incl. bank flips; every stream equals its own oracle run."""
    n = 12
    sched = [(0, 0, 37), (0, 1 if mode == 1 else 0, 291), (1, 0, 200), (1, 1 if mode == 1 else 0, 1), (2, 0, 490)]
    frames = [_synth(mode, 3, 100 + i, coherent=(i % 2 == 1)) for i in range(n)]
    seeds = [(i + 1, 1000 + i) for i in range(n)]
    enc, got = _run_device(native, device_tables, mode, 5, frames, sched,
                           [_seed_states(O, a, b) for a, b in seeds], wave=wave, prefix_sort=prefix)
    for i in range(n):
        v, exp = _oracle_run(O, oracle_tables, mode, 5, frames[i], sched, *seeds[i])
        assert (got[i] == exp).all(), "stream %d" % i
        assert (enc.get_state(native.STATE_MEM_MAIN, i) == v.memory(0)).all()
        assert (enc.get_state(native.STATE_UP_MAIN, i) == v.update_priority(0)).all()
        assert (enc.get_state(native.STATE_PACKED, i) == v.packed).all()
        cnt = enc.get_state(native.STATE_COUNTERS, i)
        assert (int(cnt[0]), int(cnt[1])) == v.draws()
    enc.close()


def test_continue_generator_and_lazy_segments(native, O, oracle_tables, device_tables):
    """restart=0 continues the same generator; an n_ops=0 restart segment is lazy
    (no side effects until the first next())."""
    import torch
    mode = 1
    frames = _synth(mode, 2, 77)
    t, s = device_tables.get(mode)
    enc = native.Encoder(mode, t, s, 1, dm=device_tables.dm[(mode, 5)])
    py, npw = _seed_states(O, 3, 4)
    enc.set_state(native.STATE_RNG_PY, py)
    enc.set_state(native.STATE_RNG_NP, npw)
    fm = torch.from_numpy(frames[None, :, 0].copy()).cuda()
    fa = torch.from_numpy(frames[None, :, 1].copy()).cuda()
    segs = [(0, 0, 1, 10), (0, 0, 0, 5), (0, 0, 0, 1), (1, 1, 1, 0), (1, 1, 0, 7), (1, 1, 0, 3)]
    got = enc.encode(fm, fa, segs).cpu().numpy()[0]
    enc.check()
    v = O.Video(mode, oracle_tables.get(mode), seed_py=3, seed_np=4)
    v.encode_frame(frames[0, 0], frames[0, 1], 0)
    a = v.next(16)
    v.encode_frame(frames[1, 0], frames[1, 1], 1)
    b = v.next(10)
    assert (got == np.concatenate([a, b])).all()
    enc.close()


def test_idempotent_when_converged(native, O, device_tables):
    """Property: once a bank is exhausted, re-encoding the same target emits only
    padding opcodes (32, target[0,0], 0,0,0,0) and changes nothing."""
    import torch
    mode = 0
    frames = _synth(mode, 1, 5)
    t, s = device_tables.get(mode)
    enc = native.Encoder(mode, t, s, 1)
    fm = torch.from_numpy(frames[None, :, 0].copy()).cuda()
    enc.encode(fm, None, [(0, 0, 1, 7000)])
    enc.check()
    assert enc.get_state(native.STATE_OUT_OF_WORK)[0] == 1
    mem = enc.get_state(native.STATE_MEM_MAIN).copy()
    ops = enc.encode(fm, None, [(0, 0, 1, 50)]).cpu().numpy()[0]
    assert (ops[:, 0] == 32).all() and (ops[:, 1] == frames[0, 0, 0, 0]).all() and (ops[:, 2:] == 0).all()
    assert (enc.get_state(native.STATE_MEM_MAIN) == mem).all()
    assert (enc.get_state(native.STATE_UP_MAIN) == 0).all()
    enc.close()


def test_reference_asserts_are_reported(native, device_tables):
    """video.py:137: a DHGR content byte with the palette bit set is an error, not silent."""
    import torch
    t, s = device_tables.get(1)
    enc = native.Encoder(1, t, s, 2)
    fm = torch.zeros((2, 1, 32, 256), dtype=torch.uint8, device="cuda")
    fa = torch.zeros((2, 1, 32, 256), dtype=torch.uint8, device="cuda")
    fm[1, 0, 3, 5] = 0x85
    enc.encode(fm, fa, [(0, 0, 1, 20)])
    with pytest.raises(AssertionError):
        enc.check()
    enc.close()
    enc = native.Encoder(1, t, s, 1)
    with pytest.raises(native.IIVError):
        enc.encode(fm[:1], fa[:1], [(0, 0, 0, 5)])  # continues a generator that does not exist
    enc.close()


@pytest.mark.parametrize("mode,wave", [(1, True), (1, False), (0, True), (1, "team"), (0, "team"), (1, "shared"), (0, "shared")])
def test_image_like_streams(native, O, oracle_tables, device_tables, mode, wave):
    """S-img input (dithered moving bars): large coherent areas, many identical
    windows, so the two best deltas tie far more often than on random data -- the wave
    kernel's nonce-resolved slow step, zero diff weights and early out-of-work all get
    exercised.  Every stream equals its own oracle run."""
    import stream_batch
    n = 6
    fm, fa = stream_batch.synth_frames_img(n, 3, mode == 1, seed=11, device="cpu")
    frames = []
    for i in range(n):
        fr = np.zeros((3, 2, 32, 256), np.uint8)
        fr[:, 0] = fm[i].numpy()
        if mode == 1:
            fr[:, 1] = fa[i].numpy()
        frames.append(fr)
    b = 1 if mode == 1 else 0
    sched = [(0, 0, 292), (0, b, 198), (1, b, 94), (1, 0, 292), (1, b, 104), (2, b, 490)]
    seeds = [(50 + i, 90 + i) for i in range(n)]
    enc, got = _run_device(native, device_tables, mode, 5, frames, sched,
                           [_seed_states(O, a, c) for a, c in seeds], wave=wave)
    for i in range(n):
        v, exp = _oracle_run(O, oracle_tables, mode, 5, frames[i], sched, *seeds[i])
        assert (got[i] == exp).all(), "stream %d" % i
        assert (enc.get_state(native.STATE_UP_MAIN, i) == v.update_priority(0)).all()
        assert (enc.get_state(native.STATE_PACKED, i) == v.packed).all()
        cnt = enc.get_state(native.STATE_COUNTERS, i)
        assert (int(cnt[0]), int(cnt[1])) == v.draws()
    enc.close()


@pytest.mark.parametrize("mode,wave", [(1, True), (0, True), (1, False), (1, "team"), (0, "team"), (1, "shared"), (0, "shared")])
def test_per_stream_schedules(native, O, oracle_tables, device_tables, mode, wave):
    """iiv_encode_streams: every stream has its own movie clock (movie.py:16-54) -- different
    clip lengths, every_n_video_frames and frame rates in ONE batch, over two calls so that
    generators are continued across calls.  Each stream equals its own oracle run."""
    import torch
    import stream_batch
    n, nf = 7, 6
    frames = [_synth(mode, nf, 300 + i, coherent=(i % 3 == 0)) for i in range(n)]
    seeds = [(11 + i, 500 + i) for i in range(n)]
    t, s = device_tables.get(mode)
    enc = native.Encoder(mode, t, s, n, dm=device_tables.dm[(mode, 5)])
    enc.set_greedy_kernel(wave)
    st = [_seed_states(O, a, b) for a, b in seeds]
    enc.set_state_all(native.STATE_RNG_PY, np.stack([x for x, _ in st]))
    enc.set_state_all(native.STATE_RNG_NP, np.stack([y for _, y in st]))
    fr = np.stack(frames)
    fm = torch.from_numpy(np.ascontiguousarray(fr[:, :, 0])).cuda()
    fa = torch.from_numpy(np.ascontiguousarray(fr[:, :, 1])).cuda() if mode == 1 else None
    clocks = [stream_batch.MovieClock(mode == 1, every_n_video_frames=1 + i % 3, input_frame_rate=(30.0, 24.0, 60.0)[i % 3])
              for i in range(n)]
    lengths = [nf - (i % 3) for i in range(n)]              # clips of 6, 5, 4 frames
    got = [[] for _ in range(n)]
    scheds = [[] for _ in range(n)]
    for part in range(2):
        sch = []
        for i in range(n):
            k = lengths[i] // 2 if part == 0 else lengths[i] - lengths[i] // 2
            sch.append(clocks[i].segments(k))
            scheds[i] += sch[-1]
        ops, totals = enc.encode_streams(fm, fa, sch)
        enc.check()
        ops = ops.cpu().numpy()
        for i in range(n):
            got[i].append(ops[i, :totals[i]])
    assert len(set(len(x) for x in scheds)) > 1              # really different schedules
    for i in range(n):
        v = O.Video(mode, oracle_tables.get(mode), seed_py=seeds[i][0], seed_np=seeds[i][1])
        exp = []
        for (f, ia, restart, k) in scheds[i]:
            if restart:
                v.encode_frame(frames[i][f, 0], frames[i][f, 1] if mode == 1 else None, ia)
            exp.append(v.next(k))
        exp = np.concatenate(exp)
        g = np.concatenate(got[i])
        assert g.shape == exp.shape and (g == exp).all(), "stream %d" % i
        assert (enc.get_state(native.STATE_UP_MAIN, i) == v.update_priority(0)).all()
        assert (enc.get_state(native.STATE_MEM_MAIN, i) == v.memory(0)).all()
    with pytest.raises(native.IIVError):
        enc.encode(fm, fa, [(0, 0, 1, 5)])                   # the streams no longer share a schedule
    enc.close()


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("kernel", ["team", True, False])
def test_generator_advanced_a_few_opcodes_per_launch(native, O, oracle_tables, device_tables, mode, kernel):
    """What video.Video does without a budget: hundreds of launches of one, two or three opcodes continuing the same
    generator (state, bitmaps, MT19937 block and window position saved and restored every time) equal one run of the oracle."""
    import torch
    t, s = device_tables.get(mode, 5)
    frames = _synth(mode, 2, 4246, coherent=True)
    fm = torch.from_numpy(frames[None, :, 0].copy()).cuda()
    fa = torch.from_numpy(frames[None, :, 1].copy()).cuda() if mode == 1 else None
    n = 420
    for chunk in (1, 2, 3):
        enc = native.Encoder(mode, t, s, 1, dm=device_tables.dm[(mode, 5)])
        enc.set_greedy_kernel(kernel)
        py, npw = _seed_states(O, 11, 12)
        enc.set_state(native.STATE_RNG_PY, py)
        enc.set_state(native.STATE_RNG_NP, npw)
        v = O.Video(mode, oracle_tables.get(mode, 5), seed_py=11, seed_np=12)
        v.encode_frame(frames[0, 0], frames[0, 1] if mode else None, 0)
        exp = v.next(n)
        got = np.concatenate([enc.encode(fm, fa, [(0, 0, 1 if i == 0 else 0, chunk)]).cpu().numpy()[0] for i in range(0, n, chunk)])
        enc.check()
        assert (got[:n] == exp).all(), (mode, kernel, chunk)
        enc.close()


@pytest.mark.parametrize("mode,fourth", [(1, False), (0, False), (1, True)])
def test_live_queue_carries_every_opcode(native, O, oracle_tables, device_tables, mode, fourth):
    """iiv_encode_live (include/iivision.h): the call is iiv_encode -- same d_ops_out, same state -- and every opcode also
    stands in the host queue, slot j = its six bytes | tag << 48; tags of earlier calls never validate a slot; both queues
    work; a launch that ends short marks the slot behind its last opcode; an encoder whose options keep it off the team
    kernel refuses before anything is launched."""
    import torch
    frames = _synth(mode, 3, 91 + mode)
    sp, sn = 21, 22
    sched = [(0, 0, 1, 292), (0, 1 if mode else 0, 1, 198), (1, 0, 1, 2000), (2, 0, 1, 7), (2, 0, 0, 300)]
    t, s = device_tables.get(mode, 5)
    enc = native.Encoder(mode, t, s, 1, dm=device_tables.dm[(mode, 5)])
    if fourth:
        enc.set_fourth_offset(True)
    fm = torch.from_numpy(np.ascontiguousarray(frames[None, :, 0])).cuda()
    fa = torch.from_numpy(np.ascontiguousarray(frames[None, :, 1])).cuda() if mode == 1 else None
    py, npw = _seed_states(O, sp, sn)
    enc.set_state_all(native.STATE_RNG_PY, py[None])
    enc.set_state_all(native.STATE_RNG_NP, npw[None])
    ov = O.Video(mode, oracle_tables.get(mode, 5), seed_py=sp, seed_np=sn)
    ov.set_fourth_offset(fourth)
    queues = [enc.live_queue(0), enc.live_queue(1)]
    assert len(queues[0]) >= 2048 and not queues[0].any() and not queues[1].any()
    ops_dev = torch.empty((1, 2048, 6), dtype=torch.uint8, device="cuda")
    for j, (f, a, r, k) in enumerate(sched):
        slot, tag = j & 1, 1000 + j
        enc.encode_live(fm, fa, (f, a, r, k), ops_dev, slot, tag)
        enc.check()
        if r:
            ov.encode_frame(frames[f, 0], frames[f, 1] if mode else None, a)
        want = ov.next(k)
        got = ops_dev[0, :k].cpu().numpy()
        assert (got == want).all(), (j, "d_ops_out")
        q = queues[slot][:k].copy()
        assert ((q >> np.uint64(48)) == tag).all(), (j, "tags")
        assert (q.view(np.uint8).reshape(k, 8)[:, :6] == want).all(), (j, "queue rows")
        if k < len(queues[slot]):
            assert int(queues[slot][k]) >> 48 != tag    # (nothing behind the last opcode)
    assert (enc.get_state(native.STATE_UP_MAIN) == ov.update_priority(0)).all()
    # a launch that ends short of its opcodes: the reference's assert at video.py:137 (DHGR: a content byte with the palette bit)
    if mode == 1 and not fourth:
        bad = frames[:1].copy()
        bad[0, 0, np.arange(64) % 32, (np.arange(64) * 37) % 120] |= 0x80    # (64 bytes: one of them is popped within the launch)
        fmb = torch.from_numpy(np.ascontiguousarray(bad[None, :, 0])).cuda()
        fab = torch.from_numpy(np.ascontiguousarray(bad[None, :, 1])).cuda()
        enc.snapshot(0)
        enc.encode_live(fmb, fab, (0, 0, 1, 2048), ops_dev, 0, 77)
        with pytest.raises(AssertionError):
            enc.check()
        q = queues[0][:2048].copy()
        valid = (q >> np.uint64(48)) == 77
        n_ok = int(valid.argmin())
        assert 0 < n_ok < 2048 and not valid[n_ok:].any()
        rows = q[:n_ok].view(np.uint8).reshape(n_ok, 8)
        assert rows[-1, 0] == 0xFF and (rows[:-1, 0] < 64).all()     # the end mark behind n_ok - 1 opcodes
        # ... which are the opcodes the exact path yields before it raises
        enc.rollback(0)
        ops = enc.encode(fmb, fab, [(0, 0, 1, n_ok - 1)])
        enc.check()
        assert (ops[0].cpu().numpy() == rows[:-1, :6]).all()
    enc.close()
    # options that keep the launches off the team kernel: refused, nothing launched, the generator bookkeeping untouched
    enc = native.Encoder(mode, t, s, 1, dm=device_tables.dm[(mode, 5)])
    enc.set_greedy_kernel("plain")
    enc.live_queue(0)
    enc.set_state_all(native.STATE_RNG_PY, py[None])
    enc.set_state_all(native.STATE_RNG_NP, npw[None])
    with pytest.raises(native.IIVError) as ei:
        enc.encode_live(fm, fa, (0, 0, 1, 50), ops_dev, 0, 5)
    assert ei.value.code == native.ERR_INVALID
    ov = O.Video(mode, oracle_tables.get(mode, 5), seed_py=sp, seed_np=sn)
    ov.set_fourth_offset(False)
    ov.encode_frame(frames[0, 0], frames[0, 1] if mode else None, 0)
    ops = enc.encode(fm, fa, [(0, 0, 1, 50)])
    enc.check()
    assert (ops[0].cpu().numpy() == ov.next(50)).all()
    enc.close()


def test_set_state_async_is_set_state_in_stream_order(native, O, oracle_tables, device_tables):
    """iiv_encoder_set_state_async (out_of_work flags, both RNG states): what iiv_encoder_set_state does, enqueued behind the
    launches already on the stream -- a generator launched after it sees the new values, one launched before does not."""
    import torch
    mode = 1
    frames = _synth(mode, 2, 17)
    t, s = device_tables.get(mode, 5)
    fm = torch.from_numpy(np.ascontiguousarray(frames[None, :, 0])).cuda()
    fa = torch.from_numpy(np.ascontiguousarray(frames[None, :, 1])).cuda()
    runs = []
    for use_async in (False, True):
        enc = native.Encoder(mode, t, s, 1, dm=device_tables.dm[(mode, 5)])
        put = enc.set_state_async if use_async else enc.set_state
        py, npw = _seed_states(O, 5, 6)
        put(native.STATE_RNG_PY, py)
        put(native.STATE_RNG_NP, npw)
        a = enc.encode(fm, fa, [(0, 0, 1, 300)]).cpu().numpy()
        py2, npw2 = _seed_states(O, 7, 8)
        for _ in range(6):      # (more calls than the staging ring has slots)
            put(native.STATE_RNG_PY, py2)
            put(native.STATE_RNG_NP, npw2)
        put(native.STATE_OUT_OF_WORK, np.array([1, 1], np.int32))
        b = enc.encode(fm, fa, [(1, 1, 1, 300)]).cpu().numpy()
        enc.check()
        runs.append((a, b, enc.get_state(native.STATE_OUT_OF_WORK).copy(), enc.get_state(native.STATE_RNG_PY).copy(),
                     enc.get_state(native.STATE_UP_MAIN).copy()))
        with pytest.raises(native.IIVError):
            enc.set_state_async(native.STATE_UP_MAIN, np.zeros((32, 256), np.int32))
        enc.close()
    for x, y in zip(*runs):
        assert (x == y).all()
    # and against the oracle: the first generator under seeds 5 / 6, the second under 7 / 8
    ov = O.Video(mode, oracle_tables.get(mode, 5), seed_py=5, seed_np=6)
    ov.encode_frame(frames[0, 0], frames[0, 1], 0)
    assert (runs[1][0][0] == ov.next(300)).all()
    L = O.lib()
    L.orc_mt_seed_py(L.orc_video_rng_py(ov._h), 7)
    L.orc_mt_seed_np(L.orc_video_rng_np(ov._h), 8)
    ov.encode_frame(frames[1, 0], frames[1, 1], 1)
    assert (runs[1][1][0] == ov.next(300)).all()
