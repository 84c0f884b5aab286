"""GPU: SURVEY 8(f4) -- a real fourth offset per opcode (IIV_OPT_FOURTH_OFFSET).  The player stores every opcode's
content byte at four offsets and video.py:146 says "Need to find 3 more offsets to fill this opcode", but the loop's exit
test `len(offsets) == 3` (video.py:180-181) counts the primary, so the reference finds two and repeats the first
(:184-186).  With the option the test reads 4.  NOT the reference's stream -- but it IS the reference's loop with one
literal changed, and that is what pins it: tests/golden/g8_fourth_offset.npz holds runs of the imported reference with
exactly that change (make_golden.py --fourth-only); the oracle equals them on the CPU (test_oracle_golden.py), and these
tests hold the kernel to both."""

import numpy as np
import pytest

from test_gpu_encode import _next_draws, _seed_states, _synth

pytestmark = pytest.mark.gpu


def _tags(g):
    return sorted(set(k.split("/")[0] for k in g.files))


def _encoder(native, device_tables, mode, pal, n, recurrence=True, prefix=True, kernel=None):
    t, s = device_tables.get(mode, pal)
    enc = native.Encoder(mode, t, s, n, dm=device_tables.dm[(mode, pal)])
    enc.set_diff_weights_mode(recurrence)
    enc.set_prefix_sort(prefix)
    enc.set_greedy_kernel(kernel)      # (the option runs in the one-wave and the eight-wave kernel; others fall back to the one-wave)
    enc.set_fourth_offset(True)
    return enc


@pytest.mark.parametrize("recurrence,prefix,kernel", [(True, True, None), (False, False, "team"), ("split", True, False), (True, False, "shared")])
def test_reference_with_exit_test_at_four(native, O, golden, device_tables, recurrence, prefix, kernel):
    """The reference's own runs with `len(offsets) == 4`: opcodes, memory maps, priorities, packed screen,
    out_of_work and both RNG positions, incl. runs to exhaustion (bag phase with three pushes per step, padding)."""
    import torch
    g8 = golden.g8_fourth_offset
    for tag in _tags(g8):
        mode, pal, sp, sn = (int(x) for x in g8[tag + "/meta"])
        frames, sched, ops = g8[tag + "/frames"], g8[tag + "/schedule"], g8[tag + "/ops"]
        enc = _encoder(native, device_tables, mode, pal, 1, recurrence, prefix, kernel)
        py, npw = _seed_states(O, sp, sn)
        enc.set_state(native.STATE_RNG_PY, py)
        enc.set_state(native.STATE_RNG_NP, npw)
        fm = torch.from_numpy(np.ascontiguousarray(frames[None, :, 0])).cuda()
        fa = torch.from_numpy(np.ascontiguousarray(frames[None, :, 1])).cuda() if mode == 1 else None
        got = enc.encode(fm, fa, [(int(f), int(a), 1, int(k)) for (f, a, k) in sched]).cpu().numpy()
        enc.check()
        bad = np.nonzero((got[0] != ops).any(axis=1))[0]
        assert len(bad) == 0, "%s: first mismatch at op %d: got %s want %s" % (tag, bad[0], got[0][bad[0]], ops[bad[0]])
        assert (enc.get_state(native.STATE_MEM_MAIN) == g8[tag + "/mem_main"]).all(), tag
        assert (enc.get_state(native.STATE_UP_MAIN) == g8[tag + "/up_main"]).all(), tag
        assert (enc.get_state(native.STATE_PACKED) == g8[tag + "/packed"]).all(), tag
        if mode == 1:
            assert (enc.get_state(native.STATE_MEM_AUX) == g8[tag + "/mem_aux"]).all(), tag
            assert (enc.get_state(native.STATE_UP_AUX) == g8[tag + "/up_aux"]).all(), tag
        assert enc.get_state(native.STATE_OUT_OF_WORK).tolist() == g8[tag + "/out_of_work"].tolist(), tag
        assert _next_draws(O, enc.get_state(native.STATE_RNG_PY), 4, True) == g8[tag + "/py_next"].tolist(), tag
        assert _next_draws(O, enc.get_state(native.STATE_RNG_NP), 4, False) == g8[tag + "/np_next"].tolist(), tag
        enc.close()


@pytest.mark.parametrize("mode,kind", [(1, "iid"), (0, "iid"), (1, "img"), (0, "img"), (1, "static"), (0, "coh")])
def test_movie_paced_batches_equal_the_oracle(native, O, oracle_tables, device_tables, mode, kind):
    """Sixteen clips per input kind through the Movie-paced batch driver (bank flips, continued generators, prefix
    selection with 4 entries per opcode): every stream equals the oracle run with the same flag.  Picture-like input
    ties on nearly every step (three winners out of ~11 equal deltas), converging input ends in the bag."""
    import stream_batch
    n, nf = 16, 6
    if kind == "img":
        fm, fa = stream_batch.synth_frames_img(n, nf, mode == 1, seed=77, device="cpu")
    else:
        fm, fa = stream_batch.synth_frames_torch(n, nf, mode == 1, seed=78, coherent=kind != "iid", device="cpu",
                                                 keep=0.98 if kind == "static" else 0.9, repeat=2 if kind == "static" else 1)
    t, s = device_tables.get(mode, 5)
    seeds = [(100 + i, 200 + i) for i in range(n)]
    b = stream_batch.StreamBatch(mode, t, s, n, seeds=seeds, dm=device_tables.dm[(mode, 5)], fourth_offset=True)
    ops, segs = b.encode_frames(fm.cuda(), fa.cuda() if fa is not None else None, nf)
    b.enc.check()
    got = ops.cpu().numpy()
    four = 0
    for i in range(n):
        v = O.Video(mode, oracle_tables.get(mode, 5), seed_py=seeds[i][0], seed_np=seeds[i][1])
        v.set_fourth_offset(True)
        exp = []
        for (fr, a, restart, k) in segs:
            if restart:
                v.encode_frame(fm[i, fr].numpy(), fa[i, fr].numpy() if fa is not None else None, a)
            exp.append(v.next(k))
        exp = np.concatenate(exp)
        bad = np.nonzero((got[i] != exp).any(axis=1))[0]
        assert len(bad) == 0, "stream %d: first mismatch at op %d: got %s want %s" % (i, bad[0], got[i][bad[0]], exp[bad[0]])
        assert (b.enc.get_state(native.STATE_UP_MAIN, i) == v.update_priority(0)).all()
        assert (b.enc.get_state(native.STATE_MEM_MAIN, i) == v.memory(0)).all()
        if mode == 1:
            assert (b.enc.get_state(native.STATE_UP_AUX, i) == v.update_priority(1)).all()
            assert (b.enc.get_state(native.STATE_MEM_AUX, i) == v.memory(1)).all()
        cnt = b.enc.get_state(native.STATE_COUNTERS, i)
        assert (int(cnt[0]), int(cnt[1])) == v.draws()
        four += int((exp[:, 5] != exp[:, 2]).sum())
    b.close()
    assert four > 0     # fourth offsets that are not a copy of the first


@pytest.mark.parametrize("mode", [1, 0])
def test_fourth_offset_leaves_less_error_per_opcode(native, O, device_tables, mode):
    """What the option is for: the same clips, the same opcode budget (three Movie-paced frames), a third more useful
    stores -- every one of 8 clips ends with less perceptual error between screen and target."""
    from test_gpu_joint import _screen_error
    import torch
    n = 8
    sched = ([(0, 0, 292), (0, 1, 198), (1, 1, 94), (1, 0, 292), (1, 1, 104), (2, 1, 188), (2, 0, 292), (2, 1, 10)]
             if mode == 1 else [(0, 0, 490), (1, 0, 490), (2, 0, 490)])
    frames = [_synth(mode, 3, 900 + i, coherent=(i % 2 == 1)) for i in range(n)]
    res = {}
    for fourth in (False, True):
        t, s = device_tables.get(mode, 5)
        enc = native.Encoder(mode, t, s, n, dm=device_tables.dm[(mode, 5)])
        enc.set_fourth_offset(fourth)
        fr = np.stack(frames)
        fm = torch.from_numpy(np.ascontiguousarray(fr[:, :, 0])).cuda()
        fa = torch.from_numpy(np.ascontiguousarray(fr[:, :, 1])).cuda() if mode == 1 else None
        st = [_seed_states(O, i + 1, i + 1) for i in range(n)]
        enc.set_state_all(native.STATE_RNG_PY, np.stack([py for py, _ in st]))
        enc.set_state_all(native.STATE_RNG_NP, np.stack([npw for _, npw in st]))
        enc.encode(fm, fa, [(int(f), int(a), 1, int(k)) for (f, a, k) in sched])
        enc.check()
        res[fourth] = _screen_error(native, device_tables, mode, enc, frames, 2, n)
        enc.close()
    assert (res[True] < res[False]).all(), (res[True], res[False])
    gain = 1.0 - res[True].sum() / res[False].sum()
    assert 0.005 < gain < 0.3, gain


@pytest.mark.parametrize("mode", [1, 0])
@pytest.mark.parametrize("fourth", [False, True])
def test_team_kernel_to_exhaustion_on_picture_like_input(native, O, oracle_tables, device_tables, mode, fourth):
    """The eight-wave kernel (one clip alone: what a single video.Video uses), with and without the option, on
    picture-like input (nearly every step is decided by the nonces) through the list, the re-queued bag -- wave 0 alone,
    ties resolved there too -- and out of work: every opcode equals the oracle's."""
    import stream_batch
    n, nf = 4, 2
    fm, fa = stream_batch.synth_frames_img(n, nf, mode == 1, seed=31, device="cpu")
    sched = [(0, 0, 1, 2500), (0, 0, 0, 2500), (1, 0, 1, 3000), (1, 0, 0, 3000)]
    enc = _encoder(native, device_tables, mode, 5, n, kernel="team")
    enc.set_fourth_offset(fourth)
    seeds = [(i + 3, i + 9) for i in range(n)]
    for i, (sp, sn) in enumerate(seeds):
        enc.set_state(native.STATE_RNG_PY, O.mt_seed_py(sp).state_words(), i)
        enc.set_state(native.STATE_RNG_NP, O.mt_seed_np(sn).state_words(), i)
    got = enc.encode(fm.cuda(), fa.cuda() if fa is not None else None, sched).cpu().numpy()
    enc.check()
    for i in range(n):
        v = O.Video(mode, oracle_tables.get(mode, 5), seed_py=seeds[i][0], seed_np=seeds[i][1])
        v.set_fourth_offset(fourth)
        exp = []
        for (fr, a, restart, k) in sched:
            if restart:
                v.encode_frame(fm[i, fr].numpy(), fa[i, fr].numpy() if fa is not None else None, a)
            exp.append(v.next(k))
        exp = np.concatenate(exp)
        bad = np.nonzero((got[i] != exp).any(axis=1))[0]
        assert len(bad) == 0, "stream %d: first mismatch at op %d: got %s want %s" % (i, bad[0], got[i][bad[0]], exp[bad[0]])
        assert v.out_of_work(0) and enc.get_state(native.STATE_OUT_OF_WORK, i)[0] == 1
        assert (enc.get_state(native.STATE_UP_MAIN, i) == v.update_priority(0)).all()
    enc.close()


def test_option_rules(native, device_tables):
    """Off unless asked for; needs the split store table (dm); together with the joint content choice since round 6, in
    either order (tests/test_gpu_joint.py holds the pair to the oracle's definition)."""
    t, s = device_tables.get(1, 5)
    enc = native.Encoder(1, t, s, 1, dm=device_tables.dm[(1, 5)])
    enc.set_fourth_offset(True)
    enc.set_content_choice(True)
    enc.set_fourth_offset(False)
    enc.set_content_choice(True)
    enc.set_fourth_offset(True)
    enc.close()
    enc = native.Encoder(1, t, s, 1)          # no dm: workgroup kernel only
    with pytest.raises(native.IIVError):
        enc.set_fourth_offset(True)
    enc.close()


def test_dropin_video_fourth_offset(O, oracle_tables):
    """The Python mirror's Video(fourth_offset=True): the lazy generator yields the oracle's stream with the flag."""
    import contextlib
    import io
    import random
    import palette
    import screen
    import video
    import video_mode

    class FG:
        input_frame_rate = 30.0

    mode = 1
    frames = _synth(mode, 2, 51)
    random.seed(4)
    np.random.seed(4)
    v = video.Video(FG(), 14700.0, mode=video_mode.VideoMode.DHGR, palette=palette.Palette.NTSC, fourth_offset=True)
    got = []
    with contextlib.redirect_stdout(io.StringIO()):
        for (fi, ia, k) in [(0, False, 120), (0, True, 77), (1, True, 140)]:
            tgt = screen.DHGRBitmap(palette.Palette.NTSC, screen.MemoryMap(1, frames[fi, 0].copy()), screen.MemoryMap(1, frames[fi, 1].copy()))
            gen = v.encode_frame(tgt, ia)
            for _ in range(k):
                page, content, offs = next(gen)
                got.append([page, content] + list(offs))
    o = O.Video(mode, oracle_tables.get(mode, 5), seed_py=4, seed_np=4)
    o.set_fourth_offset(True)
    exp = []
    for (fi, ia, k) in [(0, 0, 120), (0, 1, 77), (1, 1, 140)]:
        o.encode_frame(frames[fi, 0], frames[fi, 1], ia)
        exp.append(o.next(k))
    assert (np.array(got, np.uint8) == np.concatenate(exp)).all()
    assert (v.update_priority == o.update_priority(0).reshape(32, 256)).all()
