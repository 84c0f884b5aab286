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


def implied_diff_matrix(bitmap_cls: Type[screen.Bitmap], lower: np.ndarray, samples: int = 60000, seed: int = 0):
    """The 16x16 substitution-cost matrix a table was built from, recovered from its values alone.

    A pair of equal-length pixel strings with no adjacent pair swapped (a[k] = b[k+1], a[k+1] = b[k]) has no
    transposition to use, inserts and deletes cost 1e5, so its distance is the plain sum of the substitution
    costs of its differing pixels (make_data_tables.py:92-108): every such table entry is one linear equation
    in the 120 unknowns sub[u][v], u > v.  `lower` is the array a reference .npz holds (lower triangle).
    Returns (matrix int64 (16,16) symmetric with -1 where no sampled pair involves the colour pair,
    residual = max |equation error|: 0 means the table is consistent with the cost model)."""
    bits = int(bitmap_cls.MASKED_BITS)
    _, pix = native.pixel_strings(bitmap_cls.MODE)
    pix = pix.cpu().numpy()                                   # (n_offsets, 2^bits, n_pixels)
    rng = np.random.default_rng(seed)
    n_off, n = pix.shape[0], 1 << bits
    o = rng.integers(0, n_off, samples)
    i = rng.integers(1, n, samples)
    # neighbours in Hamming distance give short sums (well-conditioned); a few far pairs reach every colour pair
    flips = np.where(rng.random(samples) < 0.7, 1 << rng.integers(0, bits, samples), rng.integers(1, n, samples))
    j = i ^ flips
    i, j = np.maximum(i, j), np.minimum(i, j)
    keep = i != j
    o, i, j = o[keep], i[keep], j[keep]
    a, b = pix[o, i].astype(np.int64), pix[o, j].astype(np.int64)
    swap = ((a[:, :-1] == b[:, 1:]) & (a[:, 1:] == b[:, :-1]) & (a[:, :-1] != a[:, 1:])).any(axis=1)
    o, i, j, a, b = o[~swap], i[~swap], j[~swap], a[~swap], b[~swap]
    hi, lo = np.maximum(a, b), np.minimum(a, b)
    pair = np.where(hi != lo, hi * 16 + lo, -1)               # unknown index per pixel; -1: equal pixels cost 0
    m = len(o)
    A = np.zeros((m, 256), dtype=np.float64)
    rows = np.repeat(np.arange(m), pair.shape[1])
    ok = pair.reshape(-1) >= 0
    np.add.at(A, (rows[ok], pair.reshape(-1)[ok]), 1.0)
    y = lower[o, (i << bits) + j].astype(np.float64)
    used = A.any(axis=0)
    sol, *_ = np.linalg.lstsq(A[:, used], y, rcond=None)
    residual = float(np.abs(A[:, used] @ np.rint(sol) - y).max()) if m else 0.0
    out = np.full(256, -1, dtype=np.int64)
    out[used] = np.rint(sol).astype(np.int64)
    out = out.reshape(16, 16)
    out = np.where(out >= 0, out, out.T)
    out[np.arange(16), np.arange(16)] = 0
    return out, residual


def verify(data_dir: str = DATA_DIR, out=print) -> int:
    """"Pin on arrival": compare the reference-format tables somebody already holds --
    <data_dir>/<MODE>_palette_<id>_edit_distance.npz as the reference's make_data_tables.py wrote them with
    the real colormath / weighted_levenshtein -- with the tables this build computes on the GPU, entry for
    entry.  Those two packages are not available where this repo was built, so the table VALUES are the one
    thing that could not be pinned to the reference; anybody with real tables closes
    that gap by running
        python make_data_tables.py --verify transcoder/data
    On a mismatch the first differing (offset, i, j) is printed with both pixel strings and both values, and
    the substitution-cost matrix the file was built from is recovered from the file itself
    (implied_diff_matrix) and printed against this build's delta-E matrix: the entries that differ are
    exactly where colormath's sRGB -> XYZ -> Lab constants differ from the restatement.  Until those are
    fixed, Bitmap.LOAD_TABLE_FILES = True makes the encoder use the files (screen.Bitmap.edit_distances).
    Returns the number of files that differ (missing files are reported and skipped)."""
    torch = native._torch()
    n_bad = 0
    for p in palette.PALETTES.values():
        dm = compute_diff_matrix(p)
        sub = np.where(np.arange(16)[:, None] >= np.arange(16)[None, :], dm, dm.T)
        for cls in (screen.HGRBitmap, screen.DHGRBitmap):
            path = "%s/%s_palette_%d_edit_distance.npz" % (data_dir, cls.NAME, p.ID.value)
            if not os.path.exists(path):
                out("%s: missing, skipped" % path)
                continue
            theirs = np.load(path)["edit_distance"]
            bits = int(cls.MASKED_BITS)
            ours = native.build_table(cls.MODE, dm, symmetric=False)
            if theirs.dtype != np.uint16 or tuple(theirs.shape) != tuple(ours.shape):
                out("%s: DIFFERENT LAYOUT: dtype %s shape %s, expected uint16 %s" % (path, theirs.dtype, theirs.shape, tuple(ours.shape)))
                n_bad += 1
                continue
            dev = torch.from_numpy(np.ascontiguousarray(theirs).view(np.int16)).cuda()
            neq = dev != ours
            n = int(neq.sum().item())
            if n == 0:
                out("%s: IDENTICAL to the GPU-built table (%d entries)" % (path, theirs.size))
                continue
            n_bad += 1
            first = int(neq.view(-1).to(torch.uint8).argmax().item())
            o, idx = divmod(first, 1 << (2 * bits))
            i, j = idx >> bits, idx & ((1 << bits) - 1)
            _, pix = native.pixel_strings(cls.MODE)
            si = pixel_string(tuple(int(v) for v in pix[o, i].cpu().numpy()))
            sj = pixel_string(tuple(int(v) for v in pix[o, j].cpu().numpy()))
            out("%s: %d of %d entries DIFFER; first at (offset %d, i %d, j %d): strings %s / %s, file %d, GPU %d"
                % (path, n, theirs.size, o, i, j, si, sj, int(theirs[o, idx]), int(table_value(ours, o, idx))))
            implied, residual = implied_diff_matrix(cls, theirs)
            if residual != 0:
                out("   the file is NOT consistent with a substitution-cost model (max equation error %g): it was not "
                    "built by make_data_tables' cost model, or is damaged" % residual)
            for u in range(16):
                for v in range(u):
                    if implied[u, v] >= 0 and implied[u, v] != sub[u, v]:
                        out("   implied dm[%d][%d] = %d (colours %s / %s), this build computes %d"
                            % (u, v, implied[u, v], pixel_char(u), pixel_char(v), sub[u, v]))
            fixed = np.where(implied >= 0, implied, sub).astype(np.int32)
            again = native.build_table(cls.MODE, fixed, symmetric=False)
            left = int((dev != again).sum().item())
            out("   rebuilt on the GPU from the implied matrix: %d entries still differ%s"
                % (left, "" if left else " -- the delta-E matrix is the whole difference"))
            del dev, ours, again
    return n_bad


def table_value(table, o: int, idx: int) -> int:
    return int(table[o, idx].item()) & 0xffff


if __name__ == "__main__":
    import sys
    if len(sys.argv) >= 2 and sys.argv[1] == "--verify":
        sys.exit(1 if verify(sys.argv[2] if len(sys.argv) > 2 else DATA_DIR) else 0)
    main()
