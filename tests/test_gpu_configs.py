"""GPU: BASELINE.json's configurations as tests -- long Movie-paced clips (configs 3 and 4: HGR and
DHGR, NTSC) and a batch of eight //gs-palette DHGR streams (config 5 puts one of them on each of 8
GPUs; the streams never exchange data, so eight of them on one GPU is the same computation) --
against the oracle: opcode streams and final state, bit for bit.  Plus the chunked byte emission the
end-to-end bench uses."""

import numpy as np
import pytest

import stream_batch

pytestmark = pytest.mark.gpu


def _frames(mode, n_streams, n_frames, seed, kinds):
    """(n_streams, n_frames, 2, 32, 256) u8: stream i is S-iid / S-coh / S-img by kinds[i % len]."""
    out = np.zeros((n_streams, n_frames, 2, 32, 256), np.uint8)
    for i in range(n_streams):
        kind = kinds[i % len(kinds)]
        if kind == "img":
            fm, fa = stream_batch.synth_frames_img(1, n_frames, mode == 1, seed=seed + i, device="cpu")
        elif kind.startswith("static"):      # S-static: 2 % of the bytes redrawn per frame; "static4": each frame shown 4 times
            fm, fa = stream_batch.synth_frames_torch(1, n_frames, mode == 1, seed=seed + i, coherent=True, device="cpu",
                                                     keep=0.98, repeat=4 if kind == "static4" else 1)
        else:
            fm, fa = stream_batch.synth_frames_torch(1, n_frames, mode == 1, seed=seed + i, coherent=kind == "coh",
                                                     device="cpu")
        out[i, :, 0] = fm[0].numpy()
        if mode == 1:
            out[i, :, 1] = fa[0].numpy()
    return out


def _run_and_compare(native, O, oracle_tables, device_tables, mode, pal, frames, seeds, step_frames, every_n=1,
                     kernel=True, stats=None):
    stats = [] if stats is None else stats      # per stream: (opcodes emitted, of which out-of-work padding)
    import torch
    n, nf = frames.shape[:2]
    t, s = device_tables.get(mode, pal)
    fm = torch.from_numpy(np.ascontiguousarray(frames[:, :, 0])).cuda()
    fa = torch.from_numpy(np.ascontiguousarray(frames[:, :, 1])).cuda() if mode == 1 else None
    b = stream_batch.StreamBatch(mode, t, s, n, seeds=seeds, dm=device_tables.dm[(mode, pal)],
                                 every_n_video_frames=every_n)
    b.enc.set_greedy_kernel(kernel)
    got, all_segs = [], []
    while b.clock.frame_number < nf:
        ops, segs = b.encode_frames(fm, fa, min(step_frames, nf - b.clock.frame_number))
        got.append(ops.cpu().numpy())
        all_segs += segs
    b.enc.check()
    got = np.concatenate(got, axis=1)
    for i in range(n):
        v = O.Video(mode, oracle_tables.get(mode, pal), seed_py=seeds[i][0], seed_np=seeds[i][1])
        exp = []
        for (f, ia, restart, k) in all_segs:
            if restart:
                v.encode_frame(frames[i, f, 0], frames[i, f, 1] if mode == 1 else None, ia)
            exp.append(v.next(k))
        exp = np.concatenate(exp)
        bad = np.nonzero((got[i] != exp).any(axis=1))[0]
        assert len(bad) == 0, "stream %d: first differing opcode %d of %d" % (i, bad[0], len(exp))
        assert (b.enc.get_state(native.STATE_MEM_MAIN, i) == v.memory(0)).all()
        assert (b.enc.get_state(native.STATE_UP_MAIN, i) == v.update_priority(0)).all()
        assert (b.enc.get_state(native.STATE_PACKED, i) == v.packed).all()
        if mode == 1:
            assert (b.enc.get_state(native.STATE_MEM_AUX, i) == v.memory(1)).all()
            assert (b.enc.get_state(native.STATE_UP_AUX, i) == v.update_priority(1)).all()
        cnt = b.enc.get_state(native.STATE_COUNTERS, i)
        assert (int(cnt[0]), int(cnt[1])) == v.draws()
        stats.append((int(cnt[2]), int(cnt[3])))
    b.close()
    return got


@pytest.mark.parametrize("mode", [1, 0])
@pytest.mark.parametrize("kernel", [True, "team", "shared"])
def test_long_movie_paced_clips(native, O, oracle_tables, device_tables, mode, kernel):
    """Configs 3 / 4: 210 frames (102 900 opcodes per stream, ~560 generators in DHGR) of an iid, a
    coherent and an image-like clip in one batch, 50 frames per iiv_encode call -- through the
    one-wave kernel and through the eight-waves-per-stream kernel a single clip gets."""
    frames = _frames(mode, 3, 210, 4000 + mode, ("iid", "coh", "img"))
    _run_and_compare(native, O, oracle_tables, device_tables, mode, 5, frames, [(21, 22), (23, 24), (25, 26)], 50,
                     kernel=kernel)


@pytest.mark.parametrize("mode", [1, 0])
def test_thousand_frame_clips(native, O, oracle_tables, device_tables, mode):
    """BASELINE configs 3 / 4 at their full length: 1000-frame Movie-paced clips (490 000 opcodes each, ~2 680
    generators and ~1 680 bank flips in DHGR), 50-frame driver steps with generators continued across calls --
    every opcode, the final screens, priorities and both RNG positions against the oracle.  The two clips go through
    the one-wave kernel of the big batches in both its forms (plain and LDS-shared) and through the eight-waves-per-clip
    team kernel."""
    import concurrent.futures
    import torch
    n, nf = 2, 1000
    fm, fa = stream_batch.synth_frames_torch(n, nf, mode == 1, seed=77 + mode, device="cpu")
    fmh, fah = fm.numpy(), (fa.numpy() if fa is not None else None)
    seeds = [(i + 1, 100 + i) for i in range(n)]
    t, s = device_tables.get(mode, 5)
    otab = oracle_tables.get(mode, 5)
    fmd, fad = fm.cuda(), (fa.cuda() if fa is not None else None)
    runs = {}
    for kernel in ("plain", "shared", "team"):   # both one-wave forms (picture-like batches are dispatched to the plain one) and the team kernel
        b = stream_batch.StreamBatch(mode, t, s, n, seeds=seeds, dm=device_tables.dm[(mode, 5)])
        b.enc.set_greedy_kernel(kernel)
        got, segs = [], []
        for start in range(0, nf, 50):
            ops, sg = b.encode_frames(fmd, fad, 50)
            got.append(ops.cpu().numpy())
            segs += sg
        b.enc.check()
        runs[kernel] = (b, np.concatenate(got, axis=1), segs)
    segs = runs["team"][2]
    assert all(r[2] == segs for r in runs.values())

    def run(i):
        v = O.Video(mode, otab, seed_py=seeds[i][0], seed_np=seeds[i][1])
        out = []
        for (fr, ia, restart, k) in segs:
            if restart:
                v.encode_frame(fmh[i, fr], fah[i, fr] if fah is not None else None, ia)
            if k:
                out.append(v.next(k))
        return v, np.concatenate(out)

    with concurrent.futures.ThreadPoolExecutor(n) as ex:     # (the oracle's C calls release the GIL)
        res = list(ex.map(run, range(n)))
    for kernel, (b, got, _) in runs.items():
        for i, (v, exp) in enumerate(res):
            assert got[i].shape == exp.shape and exp.shape[0] >= nf * 489
            bad = np.nonzero((got[i] != exp).any(axis=1))[0]
            assert len(bad) == 0, "%s kernel, clip %d: first differing opcode %d of %d" % (kernel, i, bad[0], len(exp))
            assert (b.enc.get_state(native.STATE_MEM_MAIN, i) == v.memory(0)).all()
            assert (b.enc.get_state(native.STATE_UP_MAIN, i) == v.update_priority(0)).all()
            if mode == 1:
                assert (b.enc.get_state(native.STATE_MEM_AUX, i) == v.memory(1)).all()
                assert (b.enc.get_state(native.STATE_UP_AUX, i) == v.update_priority(1)).all()
            cnt = b.enc.get_state(native.STATE_COUNTERS, i)
            assert (int(cnt[0]), int(cnt[1])) == v.draws()
        b.close()


@pytest.mark.parametrize("mode", [1, 0])
@pytest.mark.parametrize("kernel", [True, "shared", "team", False])
def test_converging_content(native, O, oracle_tables, device_tables, mode, kernel):
    """S-static: content that converges (reference README.md:39: real video redraws the whole
    screen 7.5-8 times a second, i.e. its work list does run dry).  Every frame is the previous one with 2 % of its
    bytes redrawn (clip 1: each drawn frame shown four times), so after the first frames a generator goes through
    its whole sorted list, then through the re-queued bag (video.py:124-131, 170-178), and ends out of work with
    padding opcodes (video.py:189, 249-251) -- opcode streams and final state against the oracle."""
    frames = _frames(mode, 3, 60, 4300 + mode, ("static", "static4", "static"))
    stats = []
    _run_and_compare(native, O, oracle_tables, device_tables, mode, 5, frames, [(51, 52), (53, 54), (55, 56)], 20,
                     kernel=kernel, stats=stats)
    for ops, pads in stats:
        assert 0 < pads < ops      # the streams did run out of work, after real opcodes


def test_main_py_defaults_long_clip(native, O, oracle_tables, device_tables):
    """main.py's own defaults: DHGR, NTSC, every_n_video_frames = 2 (980 opcodes per encoded frame)."""
    frames = _frames(1, 2, 120, 4100, ("coh", "iid"))
    _run_and_compare(native, O, oracle_tables, device_tables, 1, 5, frames, [(31, 32), (33, 34)], 25, every_n=2)


def test_eight_iigs_dhgr_streams(native, O, oracle_tables, device_tables):
    """Config 5: eight independent DHGR streams on the //gs RGB palette (palette.py:33-55)."""
    frames = _frames(1, 8, 40, 4200, ("iid", "coh"))
    _run_and_compare(native, O, oracle_tables, device_tables, 1, 0, frames, [(40 + i, 60 + i) for i in range(8)], 20)


@pytest.mark.parametrize("mode", [1, 0])
def test_emit_chunks_equal_whole_stream(native, O, golden, mode):
    """iiv_emit_chunk over consecutive slices (cut at arbitrary opcodes, also right before / after an
    ACK) writes exactly the bytes iiv_emit_stream (pinned to the reference's Movie.emit_stream by
    g6_a2m.npz) writes for the whole stream."""
    import torch
    import a2m
    g = golden.g6_a2m
    addr = a2m.OpcodeAddresses(g["tick_addr"], g["special_addr"][0], g["special_addr"][1], g["special_addr"][2])
    rng = np.random.default_rng(5)
    S, n = 5, 2000
    ops = rng.integers(0, 256, (S, n, 6)).astype(np.uint8)
    ops[:, :, 0] = rng.integers(32, 64, (S, n))
    ticks = (rng.integers(0, 32, (S, n)) * 2 + 4).astype(np.uint8)
    d_ops, d_ticks = torch.from_numpy(ops).cuda(), torch.from_numpy(ticks).cuda()
    whole = a2m.emit_stream(mode, d_ops, d_ticks, addr).cpu().numpy()
    d_addr = torch.from_numpy(addr.tick.astype(np.uint16).view(np.int16).reshape(-1).copy()).cuda()
    d_err = torch.zeros(1, dtype=torch.int32, device="cuda")
    cuts = [0, 1, 290, 291, 292, 583, 584, 1000, 1459, 2000]
    pos = 0
    for a, b in zip(cuts[:-1], cuts[1:]):
        b0, nb = native.emit_chunk_range(mode, a, b - a)
        assert b0 == pos
        out = torch.full((S, nb + 3), 0xEE, dtype=torch.uint8, device="cuda")
        r0, rn = native.emit_chunk(mode, d_ops[:, a:b], a, d_addr, addr.ack, out, ticks=d_ticks[:, a:b], d_err=d_err)
        assert (r0, rn) == (b0, nb)
        o = out.cpu().numpy()
        assert (o[:, :nb] == whole[:, b0:b0 + nb]).all(), (a, b)
        assert (o[:, nb:] == 0xEE).all()
        pos += nb
    assert int(d_err.item()) == 0
    # a constant tick instead of a tick array
    out = torch.zeros((S, native.emit_chunk_range(mode, 0, n)[1]), dtype=torch.uint8, device="cuda")
    native.emit_chunk(mode, d_ops, 0, d_addr, addr.ack, out, const_tick=34)
    want = a2m.emit_stream(mode, d_ops, torch.full((S, n), 34, dtype=torch.uint8, device="cuda"), addr).cpu().numpy()
    assert (out.cpu().numpy() == want[:, :out.shape[1]]).all()


def test_emit_rejects_impossible_opcodes(native, golden):
    """a tick outside 4..66 (or odd) or a page byte outside 32..63 names no player opcode:
    the kernel must not index outside the address table; the call reports IIV_ERR_INVALID."""
    import torch
    import a2m
    g = golden.g6_a2m
    addr = a2m.OpcodeAddresses(g["tick_addr"], g["special_addr"][0], g["special_addr"][1], g["special_addr"][2])
    ops = torch.zeros((1, 4, 6), dtype=torch.uint8, device="cuda")
    ops[:, :, 0] = 40
    good = torch.full((1, 4), 34, dtype=torch.uint8, device="cuda")
    a2m.emit_stream(1, ops, good, addr)
    for bad_tick in (0, 255, 35, 68):
        t = good.clone()
        t[0, 2] = bad_tick
        with pytest.raises(native.IIVError):
            a2m.emit_stream(1, ops, t, addr)
    bad_ops = ops.clone()
    bad_ops[0, 1, 0] = 7
    with pytest.raises(native.IIVError):
        a2m.emit_stream(1, bad_ops, good, addr)
