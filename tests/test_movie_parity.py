"""f1: the batched driver against the REFERENCE's own Movie.encode /
Movie.emit_stream (movie.py:56-161), recorded by tests/golden/make_golden.py:make_movie_golden
into g7_movie.npz -- 30-frame HGR and DHGR clips, every_n_video_frames 1 and 2 (main.py's
default), the clip-end StopIteration, and audio that runs out first, on the //gs palette.

CPU: oracle video + MovieClock + oracle emit == the reference's byte stream, final state and
RNG positions (this pins the oracle and the clock to movie.py).  GPU: StreamBatch +
iiv_emit_stream reproduce the same bytes, in one call and split over several calls."""

import ctypes as C

import numpy as np
import pytest

import stream_batch

TICK_SILENCE = 34   # au = 0 -> tick 34 (movie.py:104-107)


def _tags(g):
    return sorted(set(k.split("/")[0] for k in g.files if "/" in k))


def _addresses(golden):
    import a2m
    g = golden.g6_a2m
    return a2m.OpcodeAddresses(g["tick_addr"], g["special_addr"][0], g["special_addr"][1], g["special_addr"][2])


def _next_draws(O, words, n, high):
    m = O.MT()
    m.set_state_words(words)
    L = O.lib()
    f = L.orc_py_getrandbits8 if high else L.orc_np_randint256
    return [f(C.byref(m)) for _ in range(n)]


def test_oracle_reproduces_reference_movie(O, golden, oracle_tables):
    g = golden.g7_movie
    addr = _addresses(golden)
    for t in _tags(g):
        mode, pal, seed, every_n, n_audio, _, _ = (int(x) for x in g[t + "/meta"])
        frames = g[t + "/frames"]
        segs = stream_batch.MovieClock(mode == 1, every_n_video_frames=every_n).segments(frames.shape[0], max_ticks=n_audio)
        v = O.Video(mode, oracle_tables.get(mode, pal), seed_py=seed, seed_np=seed)
        out = []
        for (f, ia, restart, n) in segs:
            if restart:
                v.encode_frame(frames[f, 0], frames[f, 1] if mode == 1 else None, ia)
            out.append(v.next(n))
        ops = np.concatenate(out)
        got = O.emit_stream(mode, ops, np.full(len(ops), TICK_SILENCE, np.uint8), addr.tick, addr.ack, addr.terminate)
        want = g[t + "/stream"]
        assert got.shape == want.shape and np.array_equal(got, want), t
        assert (v.memory(0) == g[t + "/mem_main"]).all() and (v.update_priority(0) == g[t + "/up_main"]).all(), t
        if mode == 1:
            assert (v.memory(1) == g[t + "/mem_aux"]).all() and (v.update_priority(1) == g[t + "/up_aux"]).all(), t
        assert _next_draws(O, v.rng_py().state_words(), 4, True) == g[t + "/py_next"].tolist(), t
        assert _next_draws(O, v.rng_np().state_words(), 4, False) == g[t + "/np_next"].tolist(), t


@pytest.mark.gpu
@pytest.mark.parametrize("split", [None, 7, 1])
def test_stream_batch_reproduces_reference_movie(native, O, golden, device_tables, split):
    """split=None: the whole clip in one iiv_encode call; 7 / 1: encode_frames() called for 7 / 1
    input frames at a time, so generators are continued across calls (restart == 0)."""
    import torch
    import a2m
    g = golden.g7_movie
    addr = _addresses(golden)
    for t in _tags(g):
        mode, pal, seed, every_n, n_audio, _, _ = (int(x) for x in g[t + "/meta"])
        frames = g[t + "/frames"]
        nf = frames.shape[0]
        if split == 1 and nf > 10:
            continue   # (the one-frame split is exercised on the short clip)
        tab, store = device_tables.get(mode, pal)
        fm = torch.from_numpy(np.ascontiguousarray(frames[None, :, 0])).cuda()
        fa = torch.from_numpy(np.ascontiguousarray(frames[None, :, 1])).cuda() if mode == 1 else None
        b = stream_batch.StreamBatch(mode, tab, store, 1, seeds=[(seed, seed)], dm=device_tables.dm[(mode, pal)],
                                     every_n_video_frames=every_n)
        parts, left = [], n_audio
        while b.clock.frame_number < nf and left > 0:
            k = nf - b.clock.frame_number if split is None else min(split, nf - b.clock.frame_number)
            before = b.clock.ticks
            ops, segs = b.encode_frames(fm, fa, k, max_ticks=left)
            left -= b.clock.ticks - before
            parts.append(ops[0].clone())
            if not segs:
                break
        b.enc.check()
        ops = torch.cat(parts)[None]
        ticks = torch.full((1, ops.shape[1]), TICK_SILENCE, dtype=torch.uint8, device="cuda")
        got = a2m.emit_stream(mode, ops, ticks, addr).cpu().numpy()[0]
        want = g[t + "/stream"]
        assert got.shape == want.shape, (t, got.shape, want.shape)
        bad = np.nonzero(got != want)[0]
        assert len(bad) == 0, "%s: first differing byte %d" % (t, bad[0])
        assert (b.enc.get_state(native.STATE_MEM_MAIN) == g[t + "/mem_main"]).all(), t
        assert (b.enc.get_state(native.STATE_UP_MAIN) == g[t + "/up_main"]).all(), t
        if mode == 1:
            assert (b.enc.get_state(native.STATE_MEM_AUX) == g[t + "/mem_aux"]).all(), t
            assert (b.enc.get_state(native.STATE_UP_AUX) == g[t + "/up_aux"]).all(), t
        assert _next_draws(O, b.enc.get_state(native.STATE_RNG_PY), 4, True) == g[t + "/py_next"].tolist(), t
        assert _next_draws(O, b.enc.get_state(native.STATE_RNG_NP), 4, False) == g[t + "/np_next"].tolist(), t
        b.close()
