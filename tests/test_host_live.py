"""CPU: the host-side logic of the drop-in Video's live hand-over and of the in-place access to `random`'s state
(ii-vision_amd/transcoder/video.py) -- no GPU: the queue is a numpy array filled by hand, the way the team kernel fills
the real one (include/iivision.h: iiv_encode_live)."""
import ctypes
import random
import types

import numpy as np
import pytest


def _video_module():
    import video
    return video


def _slot(tag, page, content, offs):
    v = int(page) | int(content) << 8
    for i, o in enumerate(offs):
        v |= int(o) << (16 + 8 * i)
    return np.uint64(v | int(tag) << 48)


def _lv(q, tag, n):
    return dict(q=q, q16=q.view(np.uint16).reshape(-1, 4)[:, 3], q8=q.view(np.uint8).reshape(-1, 8), tag=tag, n=n, event=None)


def _taker():
    video = _video_module()
    fake = types.SimpleNamespace(live_stats={"launches": 0, "takes": 0, "waits": 0, "wait_s": 0.0, "first_wait_s": 0.0},
                                 LIVE_TIMEOUT=0.2, _enc=types.SimpleNamespace(check=lambda: None))
    return lambda lv, k: video.Video._live_take(fake, lv, k)


def test_live_take_hands_out_the_tagged_prefix():
    take = _taker()
    rng = np.random.default_rng(3)
    n, tag = 40, 777
    rows = [(int(rng.integers(32, 64)), int(rng.integers(0, 128)), [int(x) for x in rng.integers(0, 256, 4)]) for _ in range(n)]
    q = np.zeros(64, dtype=np.uint64)
    for j in range(10):
        q[j] = _slot(tag, *rows[j])
    q[12] = _slot(tag, *rows[12])                     # (the waves of a round commit in any order: slot 12 before 10 and 11)
    q[20:30] = _slot(tag - 1, 33, 1, [1, 2, 3, 4])    # an earlier launch's slots never count
    items, ended = take(_lv(q, tag, n), 0)
    assert not ended and items == rows[:10]
    items, ended = take(_lv(q, tag, n), 4)
    assert not ended and items == rows[4:10]
    q[10], q[11] = _slot(tag, *rows[10]), _slot(tag, *rows[11])
    items, ended = take(_lv(q, tag, n), 10)
    assert not ended and items == rows[10:13]
    # every slot there: the whole rest in one take
    for j in range(n):
        q[j] = _slot(tag, *rows[j])
    items, ended = take(_lv(q, tag, n), 13)
    assert not ended and items == rows[13:]


def test_live_take_sees_the_end_mark_behind_the_last_opcode():
    take = _taker()
    tag, n = 9, 30
    q = np.zeros(32, dtype=np.uint64)
    for j in range(5):
        q[j] = _slot(tag, 40 + j, j, [j, j, j, j])
    q[5] = np.uint64(0xFF | tag << 48)                 # the launch ended after five opcodes
    items, ended = take(_lv(q, tag, n), 0)
    assert ended and [it[0] for it in items] == [40, 41, 42, 43, 44]
    items, ended = take(_lv(q, tag, n), 5)
    assert ended and items == []
    # ... and waits, up to its time-out, for a slot that has not arrived
    with pytest.raises(RuntimeError):
        take(_lv(np.zeros(8, dtype=np.uint64), tag, 4), 0)


def test_chunk_counts_what_its_iterator_has_handed_out():
    video = _video_module()
    c = video._Chunk(token=None, restart=1, produced=9, prev_live=None, slot=0)
    assert c.consumed() == 0
    it = c.hand_out(["a", "b", "c"])
    assert c.consumed() == 0 and next(it) == "a" and c.consumed() == 1
    assert list(it) == ["b", "c"] and c.consumed() == 3
    it = c.hand_out(["d", "e", "f", "g"])
    assert next(it) == "d" and c.consumed() == 4
    c.stop()                                           # settled underneath its generator: nothing more comes out
    assert list(it) == []


def test_random_state_in_place_equals_getstate_setstate():
    video = _video_module()
    saved = random.getstate()
    try:
        random.seed(41)
        [random.random() for _ in range(700)]          # (past a block boundary)
        want = random.getstate()
        raw = video._py_rng_raw()
        assert np.array_equal(np.frombuffer(raw, dtype=np.uint32), np.array(want[1], dtype=np.uint32))
        draws = [random.getrandbits(8) for _ in range(1500)]
        words = (ctypes.c_uint32 * 625)(*want[1])
        random.seed(1)
        random.gauss(0, 1)                              # (leaves a cached gauss_next behind: setstate(..., None) clears it)
        video._py_rng_write(words)
        assert random.getstate() == (want[0], want[1], None)
        assert [random.getrandbits(8) for _ in range(1500)] == draws
        # the fallback path gives the same
        addr, video._py_global[1] = video._py_global[1], 0
        try:
            assert video._py_rng_addr() == 0
            random.seed(2)
            video._py_rng_write(words)
            assert random.getstate()[1] == want[1] and video._py_rng_raw() == raw
        finally:
            video._py_global[1] = addr
    finally:
        random.setstate(saved)
