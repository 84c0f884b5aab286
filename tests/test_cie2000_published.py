"""Independent corroboration of the P1 arithmetic -- NOT a pin: the table
values depend on colormath 3.0.0 / weighted-levenshtein 0.2.2, which cannot be installed here.

1. The 34 CIE Lab pairs of Sharma, Wu & Dalal, "The CIEDE2000 Color-Difference Formula:
   Implementation Notes, Supplementary Test Data, and Mathematical Observations" (Color Res. Appl.
   30(1), 2005), Table 1 -- the data set every CIEDE2000 implementation is checked against; it
   exercises every hue-angle branch (mean hue across 0/360, the 180-degree discontinuities of
   delta-h', the blue-region rotation term).  colormath's delta_e_cie2000, which the oracle and the
   device function restate with its quirks (always-add-360 mean hue), agrees with the
   published values on all of them: the quirks only show where the chroma-dependent rotation term
   is negligible.
2. Knife edges: int() of the float delta-E goes into the tables (make_data_tables.py:68), so a
   palette pair whose delta-E lies within rounding noise of an integer could flip under a different
   libm.  Every pair's distance to the nearest integer is printed and asserted to exceed 1e-6 --
   the one documented exception is black <-> white, 99.99998 by colormath's low-precision sRGB
   matrix, which is 1.5e-5 BELOW 100: still far outside an ulp of pow/atan2/cos/exp."""

import numpy as np
import pytest

# (L1, a1, b1, L2, a2, b2, delta-E 2000)
SHARMA = np.array([
    (50.0000, 2.6772, -79.7751, 50.0000, 0.0000, -82.7485, 2.0425),
    (50.0000, 3.1571, -77.2803, 50.0000, 0.0000, -82.7485, 2.8615),
    (50.0000, 2.8361, -74.0200, 50.0000, 0.0000, -82.7485, 3.4412),
    (50.0000, -1.3802, -84.2814, 50.0000, 0.0000, -82.7485, 1.0000),
    (50.0000, -1.1848, -84.8006, 50.0000, 0.0000, -82.7485, 1.0000),
    (50.0000, -0.9009, -85.5211, 50.0000, 0.0000, -82.7485, 1.0000),
    (50.0000, 0.0000, 0.0000, 50.0000, -1.0000, 2.0000, 2.3669),
    (50.0000, -1.0000, 2.0000, 50.0000, 0.0000, 0.0000, 2.3669),
    (50.0000, 2.4900, -0.0010, 50.0000, -2.4900, 0.0009, 7.1792),
    (50.0000, 2.4900, -0.0010, 50.0000, -2.4900, 0.0010, 7.1792),
    (50.0000, 2.4900, -0.0010, 50.0000, -2.4900, 0.0011, 7.2195),
    (50.0000, 2.4900, -0.0010, 50.0000, -2.4900, 0.0012, 7.2195),
    (50.0000, -0.0010, 2.4900, 50.0000, 0.0009, -2.4900, 4.8045),
    (50.0000, -0.0010, 2.4900, 50.0000, 0.0010, -2.4900, 4.8045),
    (50.0000, -0.0010, 2.4900, 50.0000, 0.0011, -2.4900, 4.7461),
    (50.0000, 2.5000, 0.0000, 50.0000, 0.0000, -2.5000, 4.3065),
    (50.0000, 2.5000, 0.0000, 73.0000, 25.0000, -18.0000, 27.1492),
    (50.0000, 2.5000, 0.0000, 61.0000, -5.0000, 29.0000, 22.8977),
    (50.0000, 2.5000, 0.0000, 56.0000, -27.0000, -3.0000, 31.9030),
    (50.0000, 2.5000, 0.0000, 58.0000, 24.0000, 15.0000, 19.4535),
    (50.0000, 2.5000, 0.0000, 50.0000, 3.1736, 0.5854, 1.0000),
    (50.0000, 2.5000, 0.0000, 50.0000, 3.2972, 0.0000, 1.0000),
    (50.0000, 2.5000, 0.0000, 50.0000, 1.8634, 0.5757, 1.0000),
    (50.0000, 2.5000, 0.0000, 50.0000, 3.2592, 0.3350, 1.0000),
    (60.2574, -34.0099, 36.2677, 60.4626, -34.1751, 39.4387, 1.2644),
    (63.0109, -31.0961, -5.8663, 62.8187, -29.7946, -4.0864, 1.2630),
    (61.2901, 3.7196, -5.3901, 61.4292, 2.2480, -4.9620, 1.8731),
    (35.0831, -44.1164, 3.7933, 35.0232, -40.0716, 1.5901, 1.8645),
    (22.7233, 20.0904, -46.6940, 23.0331, 14.9730, -42.5619, 2.0373),
    (36.4612, 47.8580, 18.3852, 36.2715, 50.5065, 21.2231, 1.4146),
    (90.8027, -2.0831, 1.4410, 91.1528, -1.6435, 0.0447, 1.4441),
    (90.9257, -0.5406, -0.9208, 88.6381, -0.8985, -0.7239, 1.5381),
    (6.7747, -0.2908, -2.4247, 5.8714, -0.0985, -2.2286, 0.6377),
    (2.0776, 0.0795, -1.1350, 0.9033, -0.0636, -0.5514, 0.9082),
])


def test_oracle_delta_e_matches_published_ciede2000_data(O):
    got = O.delta_e_cie2000(SHARMA[:, 0:3], SHARMA[:, 3:6])
    assert np.abs(got - SHARMA[:, 6]).max() < 1e-4, np.abs(got - SHARMA[:, 6])
    back = O.delta_e_cie2000(SHARMA[:, 3:6], SHARMA[:, 0:3])     # the formula is symmetric
    assert np.abs(back - got).max() < 1e-9


@pytest.mark.gpu
def test_device_delta_e_matches_published_ciede2000_data(native, O):
    got = native.delta_e_cie2000(SHARMA[:, 0:3], SHARMA[:, 3:6])
    assert np.abs(got - SHARMA[:, 6]).max() < 1e-4
    assert np.abs(got - O.delta_e_cie2000(SHARMA[:, 0:3], SHARMA[:, 3:6])).max() < 1e-9


def _margins(f):
    """distance of every off-diagonal delta-E to the nearest integer"""
    i, j = np.nonzero(~np.eye(16, dtype=bool))
    d = f[i, j]
    return i, j, d, np.abs(d - np.rint(d))


@pytest.mark.parametrize("pal", [5, 0])
def test_knife_edge_margins_of_the_oracle(O, pal, capsys):
    f, dm = O.cie2000_matrix(O.PALETTE_RGB[pal])
    i, j, d, m = _margins(f)
    order = np.argsort(m)[:6]
    with capsys.disabled():
        print("\npalette %d: delta-E values closest to an integer: %s" % (
            pal, ", ".join("(%d,%d) %.7f" % (i[k], j[k], d[k]) for k in order)))
    identical = d < 1e-9                     # NTSC holds two identical greys: delta-E exactly 0
    bw = ((i == 0) & (j == 15)) | ((i == 15) & (j == 0))
    assert (m[~identical & ~bw] > 1e-6).all()
    assert (np.abs(d[bw] - 99.9999849) < 1e-6).all() and (dm[0, 15], dm[15, 0]) == (99, 99)
    assert (dm == np.floor(f).astype(np.int32)).all()          # int() truncates (make_data_tables.py:68)


@pytest.mark.gpu
@pytest.mark.parametrize("pal", [5, 0])
def test_knife_edge_margins_on_the_device(native, O, pal):
    """The device's float matrix keeps the same distance from every integer as the oracle's, and the
    ints agree: no pair sits close enough to an integer for the two libm's to disagree."""
    f, dm = native.cie2000_matrix(O.PALETTE_RGB[pal])
    fo, dmo = O.cie2000_matrix(O.PALETTE_RGB[pal])
    assert (dm == dmo).all() and np.abs(f - fo).max() < 1e-9
    i, j, d, m = _margins(f)
    bw = ((i == 0) & (j == 15)) | ((i == 15) & (j == 0))
    assert (m[(d > 1e-9) & ~bw] > 1e-6).all()
    assert (m[bw] > 1e-5).all()              # 1.5e-5 below 100: ~1e11 ulps
