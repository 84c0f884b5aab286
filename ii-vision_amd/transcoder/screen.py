"""Apple II hi-res screen model, GPU-backed.

Host-side mirror of the reference's transcoder/screen.py: same importable names,
arguments and error behaviour (MemoryMap / FlatMemoryMap :72-125, Bitmap
:128-547, HGRBitmap :550-816, DHGRBitmap :819-1007), with the array work done by
hand-written gfx950 kernels behind include/iivision.h:

    Bitmap._pack              -> iiv_pack                 (csrc/iiv_bitmap.hip)
    Bitmap.diff_weights       -> iiv_diff_weights
    Bitmap.compute_delta_page -> iiv_compute_delta_pages
    Bitmap.edit_distances     -> iiv_build_table / iiv_build_store_table
                                 (csrc/iiv_tables.hip), tables stay in HBM

Scalar helpers (byte_offset, masked_update, to_dots, apply, ...) are plain host
code: they act on one value.  There is no CPU fallback for the array paths: without
libiivision.so and a GPU they raise.
"""

import functools
import os
from typing import List, Optional, Tuple, Union

import numpy as np

import _iiv_native as native
import palette as pal

IntOrArray = Union[np.uint64, np.ndarray]


# --------------------------------------------------------------------------- geometry

def y_to_base_addr(y: int, page: int = 0) -> int:
    """Address of the first byte of screen row y (screen.py:16-25): rows interleave
    in three 64-row bands, each band in 8 groups of 8."""
    band, within = divmod(y, 64)
    group, line = divmod(within, 8)
    return 8192 * (page + 1) + 1024 * line + 128 * group + 40 * band


Y_TO_BASE_ADDR = [[y_to_base_addr(y, p) for y in range(192)] for p in (0, 1)]

PAGE_OFFSET_TO_X = np.zeros((32, 256), dtype=np.uint8)
PAGE_OFFSET_TO_Y = np.zeros((32, 256), dtype=np.uint8)
X_Y_TO_PAGE = np.zeros((192, 40), dtype=np.uint8)
X_Y_TO_OFFSET = np.zeros((192, 40), dtype=np.uint8)
SCREEN_HOLES = np.ones((32, 256), dtype=np.bool_)
ADDR_TO_COORDS = {}


def _build_geometry():
    ys, xs = np.mgrid[0:192, 0:40]
    base = np.array(Y_TO_BASE_ADDR[0])[:, None]
    addr = base + xs
    page = (addr >> 8) - 32
    off = addr & 0xff
    X_Y_TO_PAGE[...] = page
    X_Y_TO_OFFSET[...] = off
    PAGE_OFFSET_TO_X[page, off] = xs
    PAGE_OFFSET_TO_Y[page, off] = ys
    SCREEN_HOLES[page, off] = False
    for p in range(2):
        for y in range(192):
            for x in range(40):
                ADDR_TO_COORDS[Y_TO_BASE_ADDR[p][y] + x] = (p, y, x)


_build_geometry()


# --------------------------------------------------------------------------- memory maps

class FlatMemoryMap:
    """Linear 8K view of one hi-res screen page (screen.py:72-98)."""

    def __init__(self, screen_page: int, data: np.array = None):
        if screen_page not in [1, 2]:
            raise ValueError("Screen page out of bounds: %d" % screen_page)
        self.screen_page = screen_page
        self._addr_start = 8192 * screen_page
        self._addr_end = self._addr_start + 8191
        if data is None:
            data = np.zeros((8192,), dtype=np.uint8)
        elif data.shape != (8192,):
            raise ValueError("Unexpected shape: %r" % (data.shape,))
        self.data = data

    def to_memory_map(self):
        return MemoryMap(self.screen_page, self.data.reshape((32, 256)))

    def write(self, addr: int, val: int) -> None:
        if not self._addr_start <= addr <= self._addr_end:
            raise ValueError("Address out of range: 0x%04x" % addr)
        self.data[addr - self._addr_start] = val


class MemoryMap:
    """Page/offset view of one hi-res screen page (screen.py:101-125).  Aliases the
    caller's array (no copy), as the reference does."""

    def __init__(self, screen_page: int, page_offset: np.array = None):
        if screen_page not in [1, 2]:
            raise ValueError("Screen page out of bounds: %d" % screen_page)
        self.screen_page = screen_page
        self._page_start = 32 * screen_page
        if page_offset is None:
            page_offset = np.zeros((32, 256), dtype=np.uint8)
        elif page_offset.shape != (32, 256):
            raise ValueError("Unexpected shape: %r" % (page_offset.shape,))
        self.page_offset = page_offset

    def to_flat_memory_map(self) -> FlatMemoryMap:
        return FlatMemoryMap(self.screen_page, self.page_offset.reshape(8192))

    def write(self, page: int, offset: int, val: int) -> None:
        # Bitmap.apply passes page 0..31; like the reference this relies on
        # negative-index wraparound (screen.py:125)
        self.page_offset[page - self._page_start][offset] = val


# --------------------------------------------------------------------------- tables

class DeviceTable:
    """An edit-distance table resident in HBM, indexable like the (n_off, 2^(2*bits))
    uint16 array Bitmap.edit_distances() returns in the reference."""

    def __init__(self, mode, table, store, dm=None):
        self.mode = mode
        self.dm = dm          # 16x16 int CIE2000 matrix the table was built from
        self.table = table    # torch int16 storage of the u16 values, (n_off, 2^(2*bits))
        self.store = store    # store sub-table (see iiv_build_store_table)
        self.shape = tuple(table.shape)
        self.dtype = np.dtype(np.uint16)

    def __getitem__(self, key):
        got = self.table[key]
        if got.dim() == 0:
            return np.uint16(int(got.item()) & 0xffff)
        if got.numel() > (1 << 22):
            return _RowView(got)
        return got.cpu().numpy().view(np.uint16)

    def __len__(self):
        return self.shape[0]

    def numpy(self):
        return self.table.cpu().numpy().view(np.uint16)


class _RowView:
    """table[o]: still on the device; index it to fetch values."""

    def __init__(self, t):
        self._t = t
        self.shape = tuple(t.shape)

    def __getitem__(self, key):
        import torch
        if isinstance(key, np.ndarray):
            key = torch.from_numpy(key.astype(np.int64)).to(self._t.device)
        got = self._t[key]
        if got.dim() == 0:
            return np.uint16(int(got.item()) & 0xffff)
        return got.cpu().numpy().view(np.uint16)

    def __len__(self):
        return self.shape[0]


DATA_DIR = "transcoder/data"


# --------------------------------------------------------------------------- bitmaps

class Bitmap:
    """Packed bitmap representation of (D)HGR screen memory (screen.py:128-547)."""

    NAME = None  # type: str
    MODE = None  # type: int
    HEADER_BITS = None  # type: np.uint64
    BODY_BITS = None  # type: np.uint64
    FOOTER_BITS = None  # type: np.uint64
    MASKED_BITS = None  # type: np.uint64
    MASKED_DOTS = None  # type: np.uint64
    BYTE_MASKS = None  # type: List[np.uint64]
    BYTE_SHIFTS = None  # type: List[np.uint64]
    PHASES = None  # type: List[int]

    def __init__(self, palette: pal.Palette, main_memory: MemoryMap, aux_memory: Optional[MemoryMap]):
        self.palette = palette
        self.main_memory = main_memory
        self.aux_memory = aux_memory
        self.PACKED_BITS = self.HEADER_BITS + self.BODY_BITS + self.FOOTER_BITS
        self.SCREEN_BYTES = np.uint64(len(self.BYTE_MASKS))
        # `packed` is what the reference computes here (screen.py:151-152: self._pack()); this mirror computes it -- from the
        # bytes as they are NOW -- when it is first looked at: a Bitmap that is only handed to Video.encode_frame as a target
        # (frame_grabber.py:111-115 makes one per frame) is read through its memory maps by the kernels, never through
        # `packed`, and the device round trip of the packing kernel is then a tenth of the drop-in frame's time for nothing
        self._packed = None
        self._pack_src = (np.array(main_memory.page_offset, dtype=np.uint8),
                          np.array(aux_memory.page_offset, dtype=np.uint8) if aux_memory is not None else None)

    # ---- packing (device)
    @property
    def packed(self) -> np.ndarray:
        if self._packed is None:
            self._packed = native.pack(self.MODE, *self._pack_src)
            self._pack_src = None
        return self._packed

    @packed.setter
    def packed(self, value) -> None:
        self._packed, self._pack_src = value, None

    def _pack(self) -> None:
        """Pack the memory map(s) into (32,128) uint64 columns on the GPU (K4)."""
        aux = self.aux_memory.page_offset if self.aux_memory is not None else None
        self.packed = native.pack(self.MODE, self.main_memory.page_offset, aux)

    @staticmethod
    def _make_header(col: IntOrArray) -> IntOrArray:
        raise NotImplementedError

    @staticmethod
    def _make_footer(col: IntOrArray) -> IntOrArray:
        raise NotImplementedError

    @staticmethod
    def masked_update(byte_offset: int, old_value: IntOrArray, new_value: np.uint8) -> IntOrArray:
        raise NotImplementedError

    @staticmethod
    def byte_offset(page_offset: int, is_aux: bool) -> int:
        raise NotImplementedError

    @staticmethod
    def _byte_offsets(is_aux: bool) -> Tuple[int, int]:
        raise NotImplementedError

    @classmethod
    def to_dots(cls, masked_val: int, byte_offset: int) -> int:
        raise NotImplementedError

    # ---- scalar store (host: one value)
    def apply(self, page: int, offset: int, is_aux: bool, value: np.uint8) -> None:
        """Store one byte: update its packed column, the neighbouring column's
        footer/header, and the memory map (screen.py:256-293)."""
        bo = self.byte_offset(offset, is_aux)
        col = offset // 2
        row = self.packed[page]
        row[col] = self.masked_update(bo, row[col], value)
        self._fix_scalar_neighbours(page, col, bo)
        (self.aux_memory if is_aux else self.main_memory).write(page, offset, value)

    def _fix_scalar_neighbours(self, page: int, offset: int, byte_offset: int) -> None:
        row = self.packed[page]
        if byte_offset == 0 and offset > 0:
            row[offset - 1] = self._fix_column_left(row[offset - 1], row[offset])
        elif byte_offset == int(self.SCREEN_BYTES) - 1 and offset < 127:
            row[offset + 1] = self._fix_column_right(row[offset + 1], row[offset])

    def _fix_column_left(self, column_left: IntOrArray, column: IntOrArray) -> IntOrArray:
        keep = np.uint64((1 << int(self.HEADER_BITS + self.BODY_BITS)) - 1)
        return (column_left & keep) ^ self._make_footer(column)

    def _fix_column_right(self, column_right: IntOrArray, column: IntOrArray) -> IntOrArray:
        keep = np.uint64(((1 << int(self.BODY_BITS + self.FOOTER_BITS)) - 1) << int(self.HEADER_BITS))
        return (column_right & keep) ^ self._make_header(column)

    def _fix_array_neighbours(self, ary: np.ndarray, byte_offset: int) -> None:
        """In-place header/footer propagation for a whole array (screen.py:322-341)."""
        if byte_offset == 0:
            ary[...] = self._fix_column_left(ary, np.roll(ary, -1, axis=1))
        elif byte_offset == int(self.SCREEN_BYTES) - 1:
            ary[...] = self._fix_column_right(ary, np.roll(ary, 1, axis=1))

    # ---- tables (device)
    LOAD_TABLE_FILES = False   # True: load the .npz like the reference instead of rebuilding (see edit_distances)

    @classmethod
    def edit_distances(cls, palette_id: pal.Palette) -> DeviceTable:
        """The symmetric edit-distance table for this mode and palette, in HBM.

        By default the table is rebuilt on the GPU from the palette: the values are those
        make_data_tables writes, and the build takes milliseconds against ~1 minute to read
        and mirror a file.  With Bitmap.LOAD_TABLE_FILES = True the reference's behaviour is
        kept: transcoder/data/<NAME>_palette_<id>_edit_distance.npz (screen.py:347-350, relative
        to the cwd) is loaded and mirrored (screen.py:352-365, on the device) -- that is how a
        hand-made or third-party table is used.  Such a table has no diff matrix behind it, so
        the encoder gathers its diff weights from the table and uses the workgroup kernel.

        Cached like the reference's (functools.lru_cache there), but the flag is part of the key:
        setting LOAD_TABLE_FILES after a first call yields the file's table, not the cached rebuild."""
        return cls._edit_distances(palette_id, bool(cls.LOAD_TABLE_FILES))

    @classmethod
    @functools.lru_cache(None)
    def _edit_distances(cls, palette_id: pal.Palette, from_file: bool) -> DeviceTable:
        if from_file:
            data = "transcoder/data/%s_palette_%d_edit_distance.npz" % (cls.NAME, palette_id.value)
            if not os.path.exists(data):
                raise FileNotFoundError(
                    "Bitmap.LOAD_TABLE_FILES is set but %s does not exist (the path is relative to the "
                    "working directory %s, as in the reference); run make_data_tables.main() there or "
                    "clear the flag to rebuild the table on the GPU" % (data, os.getcwd()))
            dist = np.load(data)["edit_distance"]
            table, store = native.load_table(cls.MODE, dist)
            return DeviceTable(cls.MODE, table, store, None)
        import make_data_tables
        dm = make_data_tables.compute_diff_matrix(pal.PALETTES[palette_id])
        table = native.build_table(cls.MODE, dm, symmetric=True)
        store = native.build_store_table(cls.MODE, dm)
        return DeviceTable(cls.MODE, table, store, dm)

    @classmethod
    def mask_and_shift_data(cls, data: IntOrArray, byte_offset: int) -> IntOrArray:
        """Masks and shifts packed data into the MASKED_BITS range (screen.py:369-378)."""
        res = (data & cls.BYTE_MASKS[byte_offset]) >> cls.BYTE_SHIFTS[byte_offset]
        assert np.all(res <= 2 ** cls.MASKED_BITS)
        return res

    def byte_pair_difference(self, byte_offset: int, old_packed: np.uint64, content: np.uint8) -> np.uint16:
        """Effect of storing `content` within packed data (screen.py:383-398)."""
        old_pixels = self.mask_and_shift_data(np.uint64(old_packed), byte_offset)
        new_pixels = self.mask_and_shift_data(
            self.masked_update(byte_offset, np.uint64(old_packed), content), byte_offset)
        pair = (int(old_pixels) << int(self.MASKED_BITS)) + int(new_pixels)
        return self.edit_distances(self.palette)[byte_offset][pair]

    def diff_weights(self, source: "Bitmap", is_aux: bool) -> np.ndarray:
        """(32,256) int32 edit distance from `source` to this bitmap, per byte of the bank (K5)."""
        return native.diff_weights(self.MODE, self.edit_distances(self.palette).table,
                                   source.packed, self.packed, is_aux)

    def compute_delta_page(self, page: int, content: int, diff_weights: np.ndarray, is_aux: bool) -> np.ndarray:
        """Change in error from storing `content` at each offset of `page` (K7, screen.py:525-547)."""
        out = native.compute_delta_pages(self.MODE, self.edit_distances(self.palette).table, self.packed,
                                         [page], [int(content)], np.asarray(diff_weights).reshape(1, 256), is_aux)
        return out[0]


def _double(int7: int) -> int:
    """Each of bits 0..6 lights two dots; bit 6 a third (screen.py:712-739)."""
    out = 0
    for k in range(7):
        if int7 >> k & 1:
            out |= 0b11 << (2 * k)
    if int7 & 0x40:
        out |= 1 << 14
    return out


class HGRBitmap(Bitmap):
    """22-bit packed columns ffFbbbbbbbBAaaaaaaaHhh of two HGR bytes (screen.py:550-816)."""

    NAME = 'HGR'
    MODE = native.HGR
    HEADER_BITS = np.uint64(3)
    BODY_BITS = np.uint64(16)
    FOOTER_BITS = np.uint64(3)
    MASKED_BITS = np.uint64(14)
    MASKED_DOTS = np.uint64(18)
    BYTE_MASKS = [np.uint64(0x3fff), np.uint64(0x3fff << 8)]
    BYTE_SHIFTS = [np.uint64(0), np.uint64(8)]
    PHASES = [1, 3]

    def __init__(self, palette: pal.Palette, main_memory: MemoryMap):
        super(HGRBitmap, self).__init__(palette, main_memory, None)

    @staticmethod
    def _make_header(col: IntOrArray) -> IntOrArray:
        # palette bit (11) and data bits 5,6 (17,18) of the odd byte -> bits 2,1,0
        return ((col >> np.uint64(11)) & np.uint64(1)) << np.uint64(2) | ((col >> np.uint64(17)) & np.uint64(3))

    @staticmethod
    def _make_footer(col: IntOrArray) -> IntOrArray:
        # palette bit (10) and data bits 0,1 (3,4) of the even byte -> bits 19,20,21
        low = ((col >> np.uint64(10)) & np.uint64(1)) | (((col >> np.uint64(3)) & np.uint64(3)) << np.uint64(1))
        return low << np.uint64(19)

    @staticmethod
    @functools.lru_cache(None)
    def byte_offset(page_offset: int, is_aux: bool) -> int:
        assert not is_aux
        return page_offset & 1

    @staticmethod
    @functools.lru_cache(None)
    def _byte_offsets(is_aux: bool) -> Tuple[int, int]:
        assert not is_aux
        return 0, 1

    _double_pixels = staticmethod(functools.lru_cache(None)(_double))

    @classmethod
    @functools.lru_cache(None)
    def to_dots(cls, masked_val: int, byte_offset: int) -> int:
        """14-bit masked value -> 21 display dots (screen.py:743-789).  Three byte
        pieces are stamped left to right; a set palette bit delays a piece by one
        dot, and each stamp overwrites what the previous piece extended into."""
        assert (masked_val & (2 ** 14 - 1)) == masked_val
        if byte_offset == 0:
            body = (masked_val >> 3) & 0xff
        else:
            body = ((masked_val >> 4) & 0x7f) | (((masked_val >> 3) & 1) << 7)
        pieces = (
            # (7 data bits, palette bit, first dot, dots overwritten)
            ((masked_val & 0b011) << 5, (masked_val >> 2) & 1, -11, 0),
            (body & 0x7f, body >> 7, 3, 14),
            ((masked_val >> 12) & 0b11, (masked_val >> 11) & 1, 17, 4),
        )
        dots = 0
        for data, palbit, start, span in pieces:
            at = start + palbit
            lit = _double(data)
            dots &= ~(((1 << span) - 1) << at) if at >= 0 else ~0
            dots ^= lit << at if at >= 0 else lit >> -at
        return dots & (2 ** 21 - 1)

    @staticmethod
    def masked_update(byte_offset: int, old_value: IntOrArray, new_value: np.uint8) -> IntOrArray:
        """Store new_value at byte_offset of every entry (screen.py:792-816)."""
        v = int(new_value)
        if byte_offset == 0:
            field, at = v, 3
        else:
            field, at = ((v & 0x7f) << 1) | (v >> 7), 11  # palette bit next to the even byte's
        return (old_value & ~np.uint64(0xff << at)) ^ np.uint64(field << at)


class DHGRBitmap(Bitmap):
    """34-bit packed columns: 3-bit header, 4x7 dots (aux, main, aux, main), 3-bit footer
    (screen.py:819-1007)."""

    NAME = 'DHGR'
    MODE = native.DHGR
    HEADER_BITS = np.uint64(3)
    BODY_BITS = np.uint64(28)
    FOOTER_BITS = np.uint64(3)
    MASKED_BITS = np.uint64(13)
    MASKED_DOTS = np.uint64(10)
    BYTE_MASKS = [np.uint64(0x1fff << (7 * k)) for k in range(4)]
    BYTE_SHIFTS = [np.uint64(7 * k) for k in range(4)]
    PHASES = [1, 0, 3, 2]

    @staticmethod
    def _make_header(col: IntOrArray) -> IntOrArray:
        return (col >> np.uint64(28)) & np.uint64(0b111)

    @staticmethod
    def _make_footer(col: IntOrArray) -> IntOrArray:
        return (col & np.uint64(0b111 << 3)) << np.uint64(28)

    @staticmethod
    @functools.lru_cache(None)
    def byte_offset(page_offset: int, is_aux: bool) -> int:
        """aux even 0, main even 1, aux odd 2, main odd 3 (screen.py:956-969)."""
        return 2 * (page_offset & 1) + (0 if is_aux else 1)

    @staticmethod
    @functools.lru_cache(None)
    def _byte_offsets(is_aux: bool) -> Tuple[int, int]:
        return (0, 2) if is_aux else (1, 3)

    @classmethod
    def to_dots(cls, masked_val: int, byte_offset: int) -> int:
        return masked_val  # already a dot sequence (screen.py:983-990)

    @staticmethod
    def masked_update(byte_offset: int, old_value: IntOrArray, new_value: np.uint8) -> IntOrArray:
        at = 7 * byte_offset + 3
        return (old_value & ~np.uint64(0x7f << at)) ^ np.uint64((int(new_value) & 0x7f) << at)
