"""GPU: screen.Bitmap kernels (P2) through the C ABI against reference-generated vectors."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode,name", [(0, "HGR"), (1, "DHGR")])
def test_pack(native, golden, mode, name):
    g = golden.g4_bitmap_ops
    for k in ("src", "tgt"):
        got = native.pack(mode, g["%s_%s_main" % (name, k)], g["%s_%s_aux" % (name, k)])
        assert (got == g["%s_%s_packed" % (name, k)]).all()


@pytest.mark.parametrize("mode", [0, 1])
def test_pack_edges_and_batch(native, O, mode):
    """Empty screen, all-ones screen, and a ragged batch (n=3) against the oracle."""
    rng = np.random.default_rng(5)
    hi = 128 if mode == 1 else 256
    mains = np.stack([np.zeros((32, 256), np.uint8), np.full((32, 256), hi - 1, np.uint8),
                      rng.integers(0, hi, (32, 256), dtype=np.uint8)])
    auxs = mains[::-1].copy()
    got = native.pack(mode, mains, auxs)
    for i in range(3):
        assert (got[i] == O.pack(mode, mains[i], auxs[i])).all()
    blank = np.zeros((32, 256), np.uint8)
    assert (native.pack(mode, blank, blank) == 0).all()


@pytest.mark.parametrize("mode,name", [(0, "HGR"), (1, "DHGR")])
def test_diff_weights_and_delta_pages(native, golden, device_tables, mode, name):
    g = golden.g4_bitmap_ops
    t, _ = device_tables.get(mode)
    sp, tp = g[name + "_src_packed"], g[name + "_tgt_packed"]
    for ia in ((0, 1) if mode == 1 else (0,)):
        dw = native.diff_weights(mode, t, sp, tp, ia)
        assert dw.dtype == np.int32 and (dw == g["%s_dw_%d" % (name, ia)]).all()
        pages = g["%s_delta_pages_%d" % (name, ia)]
        cs = g["%s_delta_contents_%d" % (name, ia)]
        got = native.compute_delta_pages(mode, t, tp, pages, cs, dw[pages], ia)
        assert (got == g["%s_delta_%d" % (name, ia)]).all()


def test_diff_weights_identity_is_zero(native, device_tables):
    rng = np.random.default_rng(8)
    m = rng.integers(0, 128, (32, 256), dtype=np.uint8)
    a = rng.integers(0, 128, (32, 256), dtype=np.uint8)
    p = native.pack(1, m, a)
    t, _ = device_tables.get(1)
    for ia in (0, 1):
        assert (native.diff_weights(1, t, p, p, ia) == 0).all()
