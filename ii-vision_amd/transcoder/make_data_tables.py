"""Precompute the edit-distance tables -- on the GPU.

Host-side mirror of the reference's transcoder/make_data_tables.py (same function
names and file format).  The reference spends ~90 minutes here (README.md:67) in
805 M calls to weighted_levenshtein.dam_lev; the gfx950 table kernel
(csrc/iiv_tables.hip) does the same arithmetic in milliseconds, so main() is
dominated by copying 3 GiB off the device and writing it.

Files are written exactly where and how the reference's loader expects them
(screen.py:343-367): transcoder/data/<MODE>_palette_<id>_edit_distance.npz, key
'edit_distance', uint16, shape (n_offsets, 2^(2*MASKED_BITS)), LOWER TRIANGLE only.
np.savez (stored, not deflated) is used: np.load reads both, and deflating 3 GiB is
what would not fit the 10 s budget.
"""

import functools
import os
import time
from typing import Iterable, Type

import numpy as np

import _iiv_native as native
import colours
import palette
import screen

PIXEL_CHARS = "0123456789ABCDEF"
DATA_DIR = "transcoder/data"


def pixel_char(i: int) -> str:
    return PIXEL_CHARS[i]


@functools.lru_cache(None)
def pixel_string(pixels: Iterable[int]) -> str:
    return "".join(pixel_char(p) for p in pixels)


class EditDistanceParams:
    """Parameters of the Damerau-Levenshtein edit distance (make_data_tables.py:30-52)."""

    # Insertions and deletions never make sense between equal-length pixel strings
    insert_costs = np.ones(128, dtype=np.float64) * 100000
    delete_costs = np.ones(128, dtype=np.float64) * 100000
    # defined by the reference but never passed to dam_lev, whose default is also 1
    transpose_costs = np.ones((128, 128), dtype=np.float64)

    def __init__(self):
        self.substitute_costs = np.zeros((128, 128), dtype=np.float64)
        self.error_substitute_costs = np.zeros((128, 128), dtype=np.float64)
        self.diff_matrix = None  # the 16x16 int matrix the costs came from


def compute_diff_matrix(pal: Type[palette.BasePalette]):
    """CIE2000 delta-E between every pair of palette colours, truncated to int
    (make_data_tables.py:55-70); float64 on the GPU, one thread per pair."""
    _, dm = native.cie2000_matrix(pal.rgb_array())
    return dm


def compute_substitute_costs(pal: Type[palette.BasePalette]):
    """Costs for substituting one colour pixel for another (make_data_tables.py:73-89).
    As in the reference's fill loop the later write wins, so the matrix is the lower
    triangle of the diff matrix mirrored."""
    edp = EditDistanceParams()
    dm = compute_diff_matrix(pal)
    edp.diff_matrix = dm
    sym = np.where(np.arange(16)[:, None] >= np.arange(16)[None, :], dm, dm.T)
    idx = np.array([ord(c) for c in PIXEL_CHARS])
    edp.substitute_costs[np.ix_(idx, idx)] = sym
    edp.error_substitute_costs[np.ix_(idx, idx)] = 5 * sym
    return edp


def edit_distance(edp: EditDistanceParams, a: str, b: str, error: bool) -> np.float64:
    """Damerau-Levenshtein distance between two pixel strings (make_data_tables.py:92-108):
    one pair, on the host (Lowrance-Wagner with per-character costs; the table build
    itself never calls this)."""
    sub = edp.error_substitute_costs if error else edp.substitute_costs
    ins, dele, trans = edp.insert_costs, edp.delete_costs, 1.0
    la, lb = len(a), len(b)
    big = float("inf")
    d = np.full((la + 2, lb + 2), big)
    d[1, 1] = 0.0
    for i in range(1, la + 1):
        d[i + 1, 1] = d[i, 1] + dele[ord(a[i - 1])]
    for j in range(1, lb + 1):
        d[1, j + 1] = d[1, j] + ins[ord(b[j - 1])]
    last_row = {}
    for i in range(1, la + 1):
        ci = a[i - 1]
        last_col = 0
        for j in range(1, lb + 1):
            cj = b[j - 1]
            k, l = last_row.get(cj, 0), last_col
            if ci == cj:
                cost = 0.0
                last_col = j
            else:
                cost = sub[ord(ci), ord(cj)]
            best = min(d[i, j] + cost, d[i + 1, j] + ins[ord(cj)], d[i, j + 1] + dele[ord(ci)])
            if k > 0 and l > 0:
                best = min(best, d[k, l] + (d[i, 1] - d[k + 1, 1]) + trans + (d[1, j] - d[1, l + 1]))
            d[i + 1, j + 1] = best
        last_row[ci] = i
    res = d[la + 1, lb + 1]
    assert (0 <= res < 2 ** 16), res
    return res


def compute_edit_distance(edp: EditDistanceParams, bitmap_cls: Type[screen.Bitmap],
                          nominal_colours: Type[colours.NominalColours]) -> np.ndarray:
    """Edit distance between all pairs of pixel strings (make_data_tables.py:111-174).
    Returns the reference's array: lower triangle (j < i) filled, rest zero."""
    del nominal_colours  # the enum round trip is the identity on colour values
    t = native.build_table(bitmap_cls.MODE, _diff_matrix_of(edp), symmetric=False)
    return native.table_to_numpy(t)


def _diff_matrix_of(edp):
    if edp.diff_matrix is not None:
        return edp.diff_matrix
    idx = np.array([ord(c) for c in PIXEL_CHARS])
    return edp.substitute_costs[np.ix_(idx, idx)].astype(np.int32)


def make_edit_distance(pal: Type[palette.BasePalette], edp: EditDistanceParams,
                       bitmap_cls: Type[screen.Bitmap], nominal_colours: Type[colours.NominalColours]):
    """Write file containing (D)HGR edit distance matrix for a palette."""
    dist = compute_edit_distance(edp, bitmap_cls, nominal_colours)
    data = "%s/%s_palette_%d_edit_distance.npz" % (DATA_DIR, bitmap_cls.NAME, pal.ID.value)
    np.savez(data, edit_distance=dist)


def main():
    os.makedirs(DATA_DIR, mode=0o755, exist_ok=True)
    t0 = time.time()
    for p in palette.PALETTES.values():
        print("Processing palette %s" % p)
        edp = compute_substitute_costs(p)
        make_edit_distance(p, edp, screen.HGRBitmap, colours.HGRColours)
        make_edit_distance(p, edp, screen.DHGRBitmap, colours.DHGRColours)
    print("make_data_tables: %.2f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
