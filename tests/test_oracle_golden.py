"""CPU: the oracle (oracle/iiv_oracle.c) against vectors produced by the imported
reference (tests/golden/make_golden.py) and against the literals of the
reference's own unit tests.  This is what pins the oracle."""

import ctypes as C

import numpy as np
import pytest


def test_geometry(O, golden):
    g = golden.g0_geometry
    assert (O.screen_holes() == g["screen_holes"].astype(bool)).all()
    a, b = O.xy_tables()
    assert (a == g["x_y_to_page"]).all() and (b == g["x_y_to_offset"]).all()
    assert O.screen_holes().sum() == 512


@pytest.mark.parametrize("mode,name", [(0, "HGR"), (1, "DHGR")])
def test_dots_and_pixel_strings(O, golden, mode, name):
    """to_dots + dots_to_nominal_colour_pixel_values for every masked value."""
    g = golden.g2_dots_pixels
    dots, pix = g[name + "_dots"], g[name + "_pixels"]
    L = O.lib()
    for o in range(dots.shape[0]):
        got = np.array([L.orc_to_dots(mode, i, o) for i in range(dots.shape[1])], dtype=np.uint32)
        assert (got == dots[o]).all()
        step = 7  # every 7th value keeps the pure-ctypes loop short; all values are covered on the GPU side
        for i in range(0, dots.shape[1], step):
            assert (O.pixel_values(mode, i, o) == pix[o, i]).all()


@pytest.mark.parametrize("mode,name", [(0, "HGR"), (1, "DHGR")])
def test_bitmap_ops(O, golden, oracle_tables, mode, name):
    g = golden.g4_bitmap_ops
    tab = oracle_tables.get(mode)
    sp = O.pack(mode, g[name + "_src_main"], g[name + "_src_aux"])
    tp = O.pack(mode, g[name + "_tgt_main"], g[name + "_tgt_aux"])
    assert (sp == g[name + "_src_packed"]).all() and (tp == g[name + "_tgt_packed"]).all()
    for ia in ((0, 1) if mode == 1 else (0,)):
        dw = O.diff_weights(mode, tab, sp, tp, ia)
        assert (dw == g["%s_dw_%d" % (name, ia)]).all()
        pages, cs, ds = (g["%s_delta_%s_%d" % (name, k, ia)] for k in ("pages", "contents")), None, None
        pages = g["%s_delta_pages_%d" % (name, ia)]
        cs = g["%s_delta_contents_%d" % (name, ia)]
        ds = g["%s_delta_%d" % (name, ia)]
        for k in range(len(pages)):
            assert (O.compute_delta_page(mode, tab, tp, pages[k], cs[k], dw[pages[k]], ia) == ds[k]).all()
    packed, mm, am = sp.copy(), g[name + "_src_main"].copy(), g[name + "_src_aux"].copy()
    L = O.lib()
    for p, o, ia, val in g[name + "_apply_seq"]:
        L.orc_apply(mode, O._p(packed, C.c_uint64), O._p(mm, C.c_uint8), O._p(am, C.c_uint8),
                    int(p), int(o), int(ia), int(val))
    assert (packed == g[name + "_apply_packed"]).all()
    assert (mm == g[name + "_apply_main"]).all()
    if mode == 1:
        assert (am == g[name + "_apply_aux"]).all()


def _tags(g3):
    return sorted(set(k.split("/")[0] for k in g3.files))


@pytest.mark.parametrize("structured", [False, True])
def test_encode_runs(O, golden, oracle_tables, structured):
    """Seeded encode_frame runs (incl. exhaustion -> wrapped keys -> padding and
    abandoned one-op generators): opcode stream, final state and both RNG
    positions are bit-identical to the imported reference, for the literal heap
    form and for the restructured form the HIP kernels implement."""
    g3 = golden.g3_encode_runs
    L = O.lib()
    for tag in _tags(g3):
        mode, pal, sp, sn = (int(x) for x in g3[tag + "/meta"])
        frames, sched, ops = g3[tag + "/frames"], g3[tag + "/schedule"], g3[tag + "/ops"]
        v = O.Video(mode, oracle_tables.get(mode, pal), seed_py=sp, seed_np=sn)
        out = []
        for fi, ia, n in sched:
            v.encode_frame(frames[fi, 0], frames[fi, 1] if mode == 1 else None, ia)
            out.append(v.next(int(n), structured=structured))
        out = np.concatenate(out)
        assert (out == ops).all(), tag
        assert (v.memory(0) == g3[tag + "/mem_main"]).all(), tag
        assert (v.update_priority(0) == g3[tag + "/up_main"]).all(), tag
        assert (v.packed == g3[tag + "/packed"]).all(), tag
        if mode == 1:
            assert (v.memory(1) == g3[tag + "/mem_aux"]).all(), tag
            assert (v.update_priority(1) == g3[tag + "/up_aux"]).all(), tag
        assert [int(v.out_of_work(0)), int(v.out_of_work(1))] == g3[tag + "/out_of_work"].tolist(), tag
        rp, rn = v.rng_py(), v.rng_np()
        assert [L.orc_py_getrandbits8(C.byref(rp)) for _ in range(4)] == g3[tag + "/py_next"].tolist(), tag
        assert [L.orc_np_randint256(C.byref(rn)) for _ in range(4)] == g3[tag + "/np_next"].tolist(), tag


@pytest.mark.parametrize("structured", [False, True])
def test_fourth_offset_runs(O, golden, oracle_tables, structured):
    """f4, the opcode's fourth offset: the oracle with orc_video_set_fourth_offset against the imported reference run
    with its exit test `len(offsets) == 3` reading 4 (tests/golden/make_golden.py --fourth-only): opcodes, state,
    both RNG positions -- heap form and the restructured form the kernel implements."""
    g8 = golden.g8_fourth_offset
    L = O.lib()
    for tag in _tags(g8):
        mode, pal, sp, sn = (int(x) for x in g8[tag + "/meta"])
        frames, sched, ops = g8[tag + "/frames"], g8[tag + "/schedule"], g8[tag + "/ops"]
        v = O.Video(mode, oracle_tables.get(mode, pal), seed_py=sp, seed_np=sn)
        v.set_fourth_offset(True)
        out = []
        for fi, ia, n in sched:
            v.encode_frame(frames[fi, 0], frames[fi, 1] if mode == 1 else None, ia)
            out.append(v.next(int(n), structured=structured))
        out = np.concatenate(out)
        assert (out == ops).all(), tag
        assert (v.memory(0) == g8[tag + "/mem_main"]).all(), tag
        assert (v.update_priority(0) == g8[tag + "/up_main"]).all(), tag
        assert (v.packed == g8[tag + "/packed"]).all(), tag
        if mode == 1:
            assert (v.memory(1) == g8[tag + "/mem_aux"]).all(), tag
            assert (v.update_priority(1) == g8[tag + "/up_aux"]).all(), tag
        assert [int(v.out_of_work(0)), int(v.out_of_work(1))] == g8[tag + "/out_of_work"].tolist(), tag
        rp, rn = v.rng_py(), v.rng_np()
        assert [L.orc_py_getrandbits8(C.byref(rp)) for _ in range(4)] == g8[tag + "/py_next"].tolist(), tag
        assert [L.orc_np_randint256(C.byref(rn)) for _ in range(4)] == g8[tag + "/np_next"].tolist(), tag
    # and the flag does change the stream: the same inputs without it give the reference's opcodes (g3), whose
    # fourth offset is a copy of the first
    assert (ops[:, 5] != ops[:, 2]).any()


def test_rng_matches_python_and_numpy(O):
    """MT19937 seeding + draw conventions (video.py:178,265,291) against the real generators."""
    import random
    L = O.lib()
    for seed in (0, 1, 12345, 2**32 + 17):
        random.seed(seed)
        m = O.mt_seed_py(seed)
        assert [L.orc_py_getrandbits8(C.byref(m)) for _ in range(2000)] == [random.getrandbits(8) for _ in range(2000)]
    for seed in (0, 1, 99, 2**32 - 1):
        np.random.seed(seed)
        m = O.mt_seed_np(seed)
        assert [L.orc_np_randint256(C.byref(m)) for _ in range(2000)] == np.random.randint(0, 256, size=2000).tolist()
    random.seed(5)
    st = random.getstate()[1]
    m = O.mt_seed_py(5)
    assert m.state_words().tolist() == list(st)


def test_table_provisional_hashes(O, golden, dms):
    """Table VALUES are parity-unpinned (colormath / weighted-levenshtein are absent);
    this only checks the oracle reproduces the tables the golden runs were made with."""
    import hashlib
    g5 = golden.g5_tables
    for pal in (5, 0):
        assert (dms[pal] == g5["dm_i_%d" % pal]).all()
    tab = O.build_table(O.DHGR, dms[5], symmetric=False)
    assert hashlib.sha256(tab.tobytes()).digest() == g5["DHGR_5_lower_sha256"].tobytes()
    assert (tab[g5["DHGR_5_sample_o"], g5["DHGR_5_sample_idx"]] == g5["DHGR_5_sample_val"]).all()


def test_edit_distance_reduction(O, dms):
    """1-D DP == full weighted Lowrance-Wagner Damerau-Levenshtein on random and
    near-neighbour pixel strings."""
    sub = O.substitute_costs(dms[5])
    rng = np.random.default_rng(3)
    for n in (10, 18):
        for _ in range(400):
            a = rng.integers(0, 16, n).astype(np.uint8)
            b = a.copy() if rng.random() < 0.7 else rng.integers(0, 16, n).astype(np.uint8)
            for _ in range(int(rng.integers(0, 4))):
                k = int(rng.integers(0, n - 1))
                if rng.random() < 0.5:
                    b[k], b[k + 1] = b[k + 1], b[k]
                else:
                    b[k] = rng.integers(0, 16)
            assert O.edit_distance(sub, a, b) == O.dam_lev_full(sub, a, b)


def test_table_invariants(oracle_tables):
    """make_data_tables_test.py:18-95 re-expressed vectorised: symmetry, zero
    diagonal; DHGR phases 0-2: zeros only on the diagonal (IIGS palette: NTSC has
    two identical greys, so it has off-diagonal zeros -- video.py:194-198)."""
    t = oracle_tables.get(1, 0)
    for o in range(4):
        m = t[o].reshape(8192, 8192)
        assert (m == m.T).all()
        assert (np.diag(m) == 0).all()
        if o < 3:
            assert np.count_nonzero(m == 0) == 8192


def _screen_error(O, mode, table, v, main, aux):
    tp = O.pack(mode, main, aux if mode == 1 else None)
    return sum(int(O.diff_weights(mode, table, v.packed, tp, ia).sum()) for ia in ((0, 1) if mode == 1 else (0,)))


def test_joint_content_definition(O, oracle_tables):
    """the oracle's definition of the joint content choice (iiv_oracle.c:
    choose_content_joint) -- not reference behaviour, so what is checked is what the definition
    promises: both forms of the oracle agree, the flag off is the reference, and over a bank's
    292 opcodes the picture ends closer to the target.  (Step by step the promise is only in the
    step's own accounting -- the reference scores a store against the TARGET's neighbours,
    screen.py:542-545, not the screen's -- so a single joint step can leave more true error
    than the reference's: the first assertion pins that this is understood, not hidden.)"""
    mode = 1
    table = oracle_tables.get(mode, 5)
    rng = np.random.default_rng(12)
    tgt = rng.integers(0, 128, (2, 32, 256), dtype=np.uint8)
    tgt[:, :, 120:128] = 0
    tgt[:, :, 248:256] = 0

    def run(joint, k, structured=False):
        v = O.Video(mode, table, seed_py=3, seed_np=4)
        v.set_joint(joint)
        v.encode_frame(tgt[0], tgt[1], 0)
        ops = v.next(k, structured=structured)
        return v, ops

    e0 = _screen_error(O, mode, table, O.Video(mode, table), tgt[0], tgt[1])
    g1, _ = run(False, 1)
    j1, _ = run(True, 1)
    assert _screen_error(O, mode, table, j1, tgt[0], tgt[1]) < e0 and _screen_error(O, mode, table, g1, tgt[0], tgt[1]) < e0
    _, a = run(True, 40)
    _, b = run(True, 40, structured=True)
    assert (a == b).all()                              # heap form == restructured form, joint too
    _, c = run(False, 40)
    assert (a != c).any()                              # and it is not the reference's stream
    gj, _ = run(True, 292)
    gg, _ = run(False, 292)
    assert _screen_error(O, mode, table, gj, tgt[0], tgt[1]) < _screen_error(O, mode, table, gg, tgt[0], tgt[1])
