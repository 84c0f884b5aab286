"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.  See
oracle/iiv_oracle.h for the parity status (CIE2000 / Damerau-Levenshtein values
are PARITY UNPINNED; everything else is pinned by tests/golden).
"""

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

HGR = 0
DHGR = 1

# palette.py:37-78, rows indexed by HGRColours value (colours.py:18-44)
_NTSC = {
    0b0000: (0, 0, 0), 0b0001: (148, 12, 125), 0b1000: (99, 77, 0),
    0b1001: (249, 86, 29), 0b0100: (51, 111, 0), 0b0101: (126, 126, 126),
    0b1100: (67, 200, 0), 0b1101: (221, 206, 23), 0b0010: (32, 54, 212),
    0b0011: (188, 55, 255), 0b1010: (126, 126, 126), 0b1011: (255, 129, 236),
    0b0110: (7, 168, 225), 0b0111: (158, 172, 255), 0b1110: (93, 248, 133),
    0b1111: (255, 255, 255),
}
_IIGS = {
    0b0000: (0, 0, 0), 0b0001: (221, 0, 51), 0b1000: (136, 85, 34),
    0b1001: (255, 102, 0), 0b0100: (0, 119, 0), 0b0101: (85, 85, 85),
    0b1100: (0, 221, 0), 0b1101: (255, 255, 0), 0b0010: (0, 0, 153),
    0b0011: (221, 0, 221), 0b1010: (170, 170, 170), 0b1011: (255, 153, 136),
    0b0110: (34, 34, 255), 0b0111: (102, 170, 255), 0b1110: (0, 255, 153),
    0b1111: (255, 255, 255),
}
# Palette enum values (palette.py:18-23): IIGS=0, NTSC=5
PALETTE_RGB = {
    5: np.array([_NTSC[i] for i in range(16)], dtype=np.uint8),
    0: np.array([_IIGS[i] for i in range(16)], dtype=np.uint8),
}


def build(force=False):
    """Compile liboracle.so with gcc (building the checker is not using it)."""
    src = os.path.join(_HERE, "iiv_oracle.c")
    hdr = os.path.join(_HERE, "iiv_oracle.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _declare(L):
    u8p, i32p, u16p, u64p, u32p, f64p = (C.POINTER(C.c_uint8), C.POINTER(C.c_int32),
                                         C.POINTER(C.c_uint16), C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_uint32), C.POINTER(C.c_double))
    L.orc_masked_bits.restype = C.c_int
    L.orc_masked_dots.restype = C.c_int
    L.orc_num_offsets.restype = C.c_int
    L.orc_phase.restype = C.c_int
    L.orc_table_entries.restype = C.c_size_t
    L.orc_y_to_base_addr.restype = C.c_int
    L.orc_screen_holes.argtypes = [u8p]
    L.orc_xy_tables.argtypes = [u8p, u8p]
    L.orc_make_header.restype = C.c_uint64
    L.orc_make_header.argtypes = [C.c_int, C.c_uint64]
    L.orc_make_footer.restype = C.c_uint64
    L.orc_make_footer.argtypes = [C.c_int, C.c_uint64]
    L.orc_pack.argtypes = [C.c_int, u8p, u8p, u64p]
    L.orc_mask_and_shift.restype = C.c_uint64
    L.orc_mask_and_shift.argtypes = [C.c_int, C.c_uint64, C.c_int]
    L.orc_masked_update.restype = C.c_uint64
    L.orc_masked_update.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_uint8]
    L.orc_byte_offset.restype = C.c_int
    L.orc_apply.argtypes = [C.c_int, u64p, u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_uint8]
    L.orc_double_pixels.restype = C.c_uint32
    L.orc_double_pixels.argtypes = [C.c_uint32]
    L.orc_to_dots.restype = C.c_uint32
    L.orc_to_dots.argtypes = [C.c_int, C.c_uint32, C.c_int]
    L.orc_dots_to_pixel_values.argtypes = [C.c_int, C.c_uint32, C.c_int, u8p]
    L.orc_pixel_values.argtypes = [C.c_int, C.c_uint32, C.c_int, u8p]
    L.orc_rgb_to_lab.argtypes = [u8p, f64p]
    L.orc_delta_e_cie2000.restype = C.c_double
    L.orc_delta_e_cie2000.argtypes = [f64p, f64p]
    L.orc_cie2000_matrix.argtypes = [u8p, f64p, i32p]
    L.orc_substitute_costs.argtypes = [i32p, i32p]
    L.orc_edit_distance.restype = C.c_uint32
    L.orc_edit_distance.argtypes = [i32p, u8p, u8p, C.c_int]
    L.orc_dam_lev_full.restype = C.c_double
    L.orc_dam_lev_full.argtypes = [i32p, u8p, C.c_int, u8p, C.c_int]
    L.orc_build_table.argtypes = [C.c_int, i32p, u16p, C.c_int]
    L.orc_byte_pair_difference.restype = C.c_uint16
    L.orc_byte_pair_difference.argtypes = [C.c_int, u16p, C.c_int, C.c_uint64, C.c_uint8]
    L.orc_diff_weights.argtypes = [C.c_int, u16p, u64p, u64p, C.c_int, i32p]
    L.orc_compute_delta_page.argtypes = [C.c_int, u16p, u64p, C.c_int, C.c_uint8, i32p, C.c_int, i32p]
    L.orc_mt_init_genrand.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_mt_init_by_array.argtypes = [C.c_void_p, u32p, C.c_int]
    L.orc_mt_seed_py.argtypes = [C.c_void_p, C.c_uint64]
    L.orc_mt_seed_np.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_mt_next.restype = C.c_uint32
    L.orc_mt_next.argtypes = [C.c_void_p]
    L.orc_py_getrandbits8.restype = C.c_uint32
    L.orc_py_getrandbits8.argtypes = [C.c_void_p]
    L.orc_np_randint256.restype = C.c_uint32
    L.orc_np_randint256.argtypes = [C.c_void_p]
    L.orc_video_create.restype = C.c_void_p
    L.orc_video_create.argtypes = [C.c_int, u16p]
    L.orc_video_destroy.argtypes = [C.c_void_p]
    for f in ("orc_video_rng_py", "orc_video_rng_np", "orc_video_packed"):
        getattr(L, f).restype = C.c_void_p
        getattr(L, f).argtypes = [C.c_void_p]
    for f in ("orc_video_memory", "orc_video_update_priority"):
        getattr(L, f).restype = C.c_void_p
        getattr(L, f).argtypes = [C.c_void_p, C.c_int]
    L.orc_video_out_of_work.restype = C.c_int
    L.orc_video_out_of_work.argtypes = [C.c_void_p, C.c_int]
    L.orc_video_reset_out_of_work.argtypes = [C.c_void_p]
    L.orc_video_set_joint.argtypes = [C.c_void_p, C.c_int]
    L.orc_video_joint_stats.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
    L.orc_video_joint_stats.restype = None
    L.orc_video_set_fourth_offset.argtypes = [C.c_void_p, C.c_int]
    L.orc_video_encode_frame.argtypes = [C.c_void_p, u8p, u8p, C.c_int]
    L.orc_video_next.restype = C.c_int
    L.orc_video_next.argtypes = [C.c_void_p, C.c_int, u8p]
    L.orc_video_next_structured.restype = C.c_int
    L.orc_video_next_structured.argtypes = [C.c_void_p, C.c_int, u8p]
    L.orc_video_draws_py.restype = C.c_uint64
    L.orc_video_draws_py.argtypes = [C.c_void_p]
    L.orc_video_draws_np.restype = C.c_uint64
    L.orc_video_draws_np.argtypes = [C.c_void_p]
    L.orc_emit_stream.restype = C.c_size_t
    L.orc_emit_stream.argtypes = [C.c_int, C.c_int, u8p, u8p, u16p, C.c_uint16, C.c_uint16, C.c_long, u8p]


# --------------------------------------------------------------------------
# thin numpy-level helpers
# --------------------------------------------------------------------------

def masked_bits(mode):
    return lib().orc_masked_bits(mode)


def masked_dots(mode):
    return lib().orc_masked_dots(mode)


def num_offsets(mode):
    return lib().orc_num_offsets(mode)


def screen_holes():
    h = np.zeros((32, 256), dtype=np.uint8)
    lib().orc_screen_holes(_p(h, C.c_uint8))
    return h.astype(bool)


def xy_tables():
    a = np.zeros((192, 40), dtype=np.uint8)
    b = np.zeros((192, 40), dtype=np.uint8)
    lib().orc_xy_tables(_p(a, C.c_uint8), _p(b, C.c_uint8))
    return a, b


def pack(mode, main_mem, aux_mem=None):
    main_mem = np.ascontiguousarray(main_mem, dtype=np.uint8)
    if aux_mem is None:
        aux_mem = np.zeros((32, 256), dtype=np.uint8)
    aux_mem = np.ascontiguousarray(aux_mem, dtype=np.uint8)
    out = np.zeros((32, 128), dtype=np.uint64)
    lib().orc_pack(mode, _p(main_mem, C.c_uint8), _p(aux_mem, C.c_uint8), _p(out, C.c_uint64))
    return out


def pixel_values(mode, masked_val, byte_offset):
    n = masked_dots(mode)
    out = np.zeros(n, dtype=np.uint8)
    lib().orc_pixel_values(mode, int(masked_val), int(byte_offset), _p(out, C.c_uint8))
    return out


def dots_to_pixel_values(num_bits, dots, init_phase):
    out = np.zeros(num_bits, dtype=np.uint8)
    lib().orc_dots_to_pixel_values(int(num_bits), int(dots), int(init_phase), _p(out, C.c_uint8))
    return out


def cie2000_matrix(rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8).reshape(48)
    f = np.zeros((16, 16), dtype=np.float64)
    i = np.zeros((16, 16), dtype=np.int32)
    lib().orc_cie2000_matrix(_p(rgb, C.c_uint8), _p(f, C.c_double), _p(i, C.c_int32))
    return f, i


def delta_e_cie2000(lab1, lab2):
    L = lib()
    L.orc_delta_e_cie2000.restype = C.c_double
    a = np.ascontiguousarray(lab1, dtype=np.float64).reshape(-1, 3)
    b = np.ascontiguousarray(lab2, dtype=np.float64).reshape(-1, 3)
    return np.array([L.orc_delta_e_cie2000(_p(a[i], C.c_double), _p(b[i], C.c_double)) for i in range(len(a))])


def substitute_costs(dm):
    dm = np.ascontiguousarray(dm, dtype=np.int32).reshape(256)
    out = np.zeros((16, 16), dtype=np.int32)
    lib().orc_substitute_costs(_p(dm, C.c_int32), _p(out, C.c_int32))
    return out


def edit_distance(sub, a, b):
    sub = np.ascontiguousarray(sub, dtype=np.int32).reshape(256)
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    return int(lib().orc_edit_distance(_p(sub, C.c_int32), _p(a, C.c_uint8), _p(b, C.c_uint8), len(a)))


def dam_lev_full(sub, a, b):
    sub = np.ascontiguousarray(sub, dtype=np.int32).reshape(256)
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    return float(lib().orc_dam_lev_full(_p(sub, C.c_int32), _p(a, C.c_uint8), len(a),
                                        _p(b, C.c_uint8), len(b)))


def build_table(mode, dm, symmetric=True):
    """(num_offsets, 2**(2*bits)) uint16."""
    dm = np.ascontiguousarray(dm, dtype=np.int32).reshape(256)
    bits = masked_bits(mode)
    out = np.empty((num_offsets(mode), 1 << (2 * bits)), dtype=np.uint16)
    lib().orc_build_table(mode, _p(dm, C.c_int32), _p(out, C.c_uint16), 1 if symmetric else 0)
    return out


def diff_weights(mode, table, src_packed, tgt_packed, is_aux):
    src_packed = np.ascontiguousarray(src_packed, dtype=np.uint64)
    tgt_packed = np.ascontiguousarray(tgt_packed, dtype=np.uint64)
    out = np.zeros((32, 256), dtype=np.int32)
    lib().orc_diff_weights(mode, _p(table, C.c_uint16), _p(src_packed, C.c_uint64),
                           _p(tgt_packed, C.c_uint64), int(is_aux), _p(out, C.c_int32))
    return out


def compute_delta_page(mode, table, tgt_packed, page, content, dw_row, is_aux):
    tgt_packed = np.ascontiguousarray(tgt_packed, dtype=np.uint64)
    dw_row = np.ascontiguousarray(dw_row, dtype=np.int32)
    out = np.zeros(256, dtype=np.int32)
    lib().orc_compute_delta_page(mode, _p(table, C.c_uint16), _p(tgt_packed, C.c_uint64), int(page),
                                 int(content), _p(dw_row, C.c_int32), int(is_aux), _p(out, C.c_int32))
    return out


def emit_stream(mode, ops, ticks, tick_addr, ack_addr, terminate_addr, max_bytes_out=None):
    """movie.emit_stream bytes for one stream (ops (n,6) u8, ticks (n,) 4..66)."""
    ops = np.ascontiguousarray(ops, dtype=np.uint8).reshape(-1, 6)
    ticks = np.ascontiguousarray(ticks, dtype=np.uint8)
    ta = np.ascontiguousarray(tick_addr, dtype=np.uint16).reshape(1024)
    n = len(ops)
    out = np.zeros(7 + 7 * n + 4 * (n // 291 + 2) + 2 + 2048, dtype=np.uint8)
    ln = lib().orc_emit_stream(mode, n, _p(ops, C.c_uint8), _p(ticks, C.c_uint8), _p(ta, C.c_uint16),
                               int(ack_addr), int(terminate_addr), int(max_bytes_out or 0), _p(out, C.c_uint8))
    return out[:ln].copy()


DITHER_DIFFUSION = 256   # ORC_DITHER_DIFFUSION: Floyd-Steinberg error diffusion instead of the ordered dither


def frame_to_memory_map(mode, palette_rgb, rgb, dither=0):
    """(main, aux) (32,256) u8 memory maps of one 280x192 RGB frame (oracle's DEFINITION of f3)."""
    pal = np.ascontiguousarray(palette_rgb, dtype=np.uint8).reshape(48)
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8).reshape(192 * 280 * 3)
    main = np.zeros((32, 256), dtype=np.uint8)
    aux = np.zeros((32, 256), dtype=np.uint8)
    lib().orc_frame_to_memory_map(mode, _p(pal, C.c_uint8), _p(rgb, C.c_uint8), int(dither),
                                  _p(main, C.c_uint8), _p(aux, C.c_uint8))
    return main, (aux if mode == DHGR else None)


class MT(C.Structure):
    _fields_ = [("mt", C.c_uint32 * 624), ("idx", C.c_int32)]

    def state_words(self):
        """625 uint32: 624 state words + position (random.getstate()[1] layout)."""
        a = np.zeros(625, dtype=np.uint32)
        a[:624] = np.frombuffer(self.mt, dtype=np.uint32)
        a[624] = self.idx
        return a

    def set_state_words(self, a):
        a = np.asarray(a, dtype=np.uint32)
        C.memmove(self.mt, a[:624].tobytes(), 624 * 4)
        self.idx = int(a[624])


def mt_seed_py(seed):
    m = MT()
    lib().orc_mt_seed_py(C.byref(m), int(seed))
    return m


def mt_seed_np(seed):
    m = MT()
    lib().orc_mt_seed_np(C.byref(m), int(seed))
    return m


class Video:
    """orc_video wrapper: the oracle's video.Video (video.py:16-301)."""

    def __init__(self, mode, table, seed_py=None, seed_np=None):
        self.mode = mode
        self._table = np.ascontiguousarray(table, dtype=np.uint16)
        self._L = lib()
        self._h = self._L.orc_video_create(mode, _p(self._table, C.c_uint16))
        if seed_py is not None:
            self._L.orc_mt_seed_py(self._L.orc_video_rng_py(self._h), int(seed_py))
        if seed_np is not None:
            self._L.orc_mt_seed_np(self._L.orc_video_rng_np(self._h), int(seed_np))

    def __del__(self):
        try:
            if self._h:
                self._L.orc_video_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _view(self, ptr, shape, dtype):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        buf = (C.c_uint8 * n).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def memory(self, is_aux):
        return self._view(self._L.orc_video_memory(self._h, int(is_aux)), (32, 256), np.uint8)

    def update_priority(self, is_aux):
        return self._view(self._L.orc_video_update_priority(self._h, int(is_aux)), (32, 256), np.int32)

    @property
    def packed(self):
        return self._view(self._L.orc_video_packed(self._h), (32, 128), np.uint64)

    def rng_py(self):
        return MT.from_address(self._L.orc_video_rng_py(self._h))

    def rng_np(self):
        return MT.from_address(self._L.orc_video_rng_np(self._h))

    def out_of_work(self, is_aux):
        return bool(self._L.orc_video_out_of_work(self._h, int(is_aux)))

    def set_joint(self, joint):
        """f4: joint choice of the content byte (README.md:212-215); not reference behaviour."""
        self._L.orc_video_set_joint(self._h, 1 if joint else 0)

    def joint_stats(self, on=True):
        """diagnostic (iiv_oracle.c: joint_prune_stats): switch the counters on / off and read them"""
        out = (C.c_uint64 * 6)()
        self._L.orc_video_joint_stats(self._h, 1 if on else 0, out)
        return dict(zip(("steps", "eligible", "eligible_dw0", "looked_at", "behind_16", "behind_16_prunable"), [int(x) for x in out]))

    def set_fourth_offset(self, fourth):
        """f4: up to three extra offsets per opcode (video.py:181 with 4 for 3); not reference behaviour."""
        self._L.orc_video_set_fourth_offset(self._h, 1 if fourth else 0)

    def reset_out_of_work(self):
        self._L.orc_video_reset_out_of_work(self._h)

    def draws(self):
        return int(self._L.orc_video_draws_py(self._h)), int(self._L.orc_video_draws_np(self._h))

    def encode_frame(self, tgt_main, tgt_aux, is_aux):
        tm = np.ascontiguousarray(tgt_main, dtype=np.uint8)
        ta = np.ascontiguousarray(tgt_aux if tgt_aux is not None else np.zeros((32, 256), np.uint8),
                                  dtype=np.uint8)
        self._L.orc_video_encode_frame(self._h, _p(tm, C.c_uint8), _p(ta, C.c_uint8), int(is_aux))

    def next(self, k, structured=False):
        out = np.zeros((k, 6), dtype=np.uint8)
        f = self._L.orc_video_next_structured if structured else self._L.orc_video_next
        rc = f(self._h, int(k), _p(out, C.c_uint8))
        if rc != 0:
            raise AssertionError("oracle: reference assertion would fire (code %d)" % rc)
        return out
