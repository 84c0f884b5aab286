"""CPU: the two identities round 4's kernels rest on, checked against the ORACLE's colour strings and recurrence
(the oracle is the checker here; the kernels' own exhaustive checks are tests/test_gpu_tables.py).

1. For the colour strings of this machine -- sliding 4-dot windows, colours.py:100-134 -- the weighted
   Damerau-Levenshtein recurrence of make_data_tables.py:92-108 is a plain SUM of per-pixel terms
       g_k = s_k,  or  min(s_k, 1 - s_{k-1}) where pixels (k-1, k) transpose,
   because two transpositions can never overlap (a[k-1], a[k], a[k+1] = X, Y, X with X != Y cannot occur).
   What the DHGR prologue evaluates (csrc/iiv_tables.hip: dw_piece_kernel).
2. The store value S = ED(window with `content` poked, window) is L1 + RF with L1 = E[M-1] of the left half and
   RF = min(r1, r0 - s) of the right half, s = the substitution cost of pixel M-1 (csrc/iiv_stream.h: narrow form):
   the path over a transposition across the cut needs no third table and no exceptions.
Both for the shipped palettes and for random (also asymmetric) diff matrices."""

import ctypes as C

import numpy as np
import pytest

INF = 1 << 20


def _strings(O, mode, o):
    bits, n = (14, 18) if mode == O.HGR else (13, 10)
    out = np.zeros((1 << bits, n), np.uint8)
    buf = np.zeros(n, np.uint8)
    L = O.lib()
    for w in range(1 << bits):
        L.orc_pixel_values(mode, w, o, O._p(buf, C.c_uint8))
        out[w] = buf
    return out.astype(np.int64)


def _chain(sub, a, b, k0, k1, e2, e1):
    for k in range(k0, k1):
        e = e1 + sub[a[..., k], b[..., k]]
        if k >= 1:
            t = (a[..., k - 1] == b[..., k]) & (a[..., k] == b[..., k - 1])
            e = np.where(t, np.minimum(e, e2 + 1), e)
        e2, e1 = e1, e
    return e2, e1


def _matrices(O, mode):
    rng = np.random.default_rng(41 + mode)
    hi = 114 if mode == O.HGR else 205
    yield "NTSC", O.cie2000_matrix(O.PALETTE_RGB[5])[1]
    yield "IIGS", O.cie2000_matrix(O.PALETTE_RGB[0])[1]
    dm = rng.integers(0, hi, (16, 16)).astype(np.int32)
    yield "random asymmetric", dm
    dm = np.triu(dm, 1)
    dm = dm + dm.T
    dm[3, 9] = dm[9, 3] = 0          # two "identical" colours, as NTSC's two greys
    yield "random symmetric", dm


@pytest.mark.parametrize("mode_name", ["DHGR", "HGR"])
def test_distance_is_a_sum_of_per_pixel_terms(O, mode_name):
    mode = getattr(O, mode_name)
    bits, n = (14, 18) if mode == O.HGR else (13, 10)
    rng = np.random.default_rng(7)
    for o in range(2 if mode == O.HGR else 4):
        st = _strings(O, mode, o)
        i = rng.integers(0, 1 << bits, 300_000)
        far = rng.integers(0, 1 << bits, i.size)
        near = i ^ (1 << rng.integers(0, bits, i.size)) ^ np.where(rng.random(i.size) < 0.5, 1 << rng.integers(0, bits, i.size), 0)
        for name, dm in _matrices(O, mode):
            sub = dm.astype(np.int64).copy()
            np.fill_diagonal(sub, 0)
            for j in (far, near):
                a, b = st[i], st[j]
                z = np.zeros(i.size, np.int64)
                _, ref = _chain(sub, a, b, 0, n, z + INF, z)
                s = sub[a, b]
                tot = s[:, 0].copy()
                for k in range(1, n):
                    t = (a[:, k - 1] == b[:, k]) & (a[:, k] == b[:, k - 1])
                    tot += np.where(t, np.minimum(s[:, k], 1 - s[:, k - 1]), s[:, k])
                assert (tot == ref).all(), (mode_name, o, name)
        # and the structural reason: no string holds X, Y, X with X != Y
        assert not ((st[:, :-2] == st[:, 2:]) & (st[:, :-2] != st[:, 1:-1])).any()


def test_store_value_is_left_plus_folded_right(O):
    """DHGR, every (content, window) of two of the four byte offsets, two matrices (HGR and the rest: on the device)."""
    mode, bits, n, m = O.DHGR, 13, 10, 5
    win = np.arange(1 << bits)[None, :]
    c = np.arange(128)[:, None]
    tgt = np.broadcast_to(win, (128, 1 << bits))
    src = (tgt & ~(0x7f << 3)) | (c << 3)                       # masked_update (screen.py:993-1007)
    mats = list(_matrices(O, mode))
    for o, (name, dm) in ((0, mats[0]), (3, mats[2])):
        sub = dm.astype(np.int64).copy()
        np.fill_diagonal(sub, 0)
        st = _strings(O, mode, o)
        a, b = st[src], st[tgt]
        z = np.zeros(src.shape, np.int64)
        _, S = _chain(sub, a, b, 0, n, z + INF, z)
        _, l1 = _chain(sub, a, b, 0, m, z + INF, z)
        _, r0 = _chain(sub, a, b, m, n, z, z + INF)
        _, r1 = _chain(sub, a, b, m, n, z + INF, z)
        rf = np.minimum(r1, r0 - sub[a[..., m - 1], b[..., m - 1]])
        assert (l1 + rf == S).all(), (o, name)
        assert ((l1 + r1) != S).sum() == S.size // 64            # (what round 3's exception masks covered)
        # RF is a function of the right half's bits alone: window bits 4..12, content bits 1..6
        key = (((c >> 1) & 63) << 9) | (tgt >> 4)
        lo = np.full(64 * 512, INF)
        hi = np.full(64 * 512, -INF)
        np.minimum.at(lo, key.ravel(), rf.ravel())
        np.maximum.at(hi, key.ravel(), rf.ravel())
        assert (lo == hi).all(), (o, name)
