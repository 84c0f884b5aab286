"""ctypes binding of libiivision.so (include/iivision.h).

PyTorch is plumbing here: it owns device memory (torch tensors whose data_ptr()
is handed to the C ABI) and the HIP stream.  There is no CPU fallback: if the
library or a GPU is missing, every compute entry point raises.
"""

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IIV_LIB") or os.path.join(os.path.dirname(_HERE), "libiivision.so")

HGR = 0
DHGR = 1

OK = 0
ERR_INVALID, ERR_HIP, ERR_NO_DEVICE, ERR_ASSERT, ERR_OVERFLOW = -1, -2, -3, -4, -5

STATE_MEM_MAIN, STATE_MEM_AUX, STATE_UP_MAIN, STATE_UP_AUX = 0, 1, 2, 3
STATE_RNG_PY, STATE_RNG_NP, STATE_OUT_OF_WORK, STATE_PACKED, STATE_COUNTERS = 4, 5, 6, 7, 8
OPT_DIFF_WEIGHTS, DW_TABLE, DW_RECURRENCE, DW_SPLIT = 1, 0, 1, 2
OPT_GREEDY_KERNEL, GREEDY_WAVE, GREEDY_WORKGROUP, GREEDY_AUTO, GREEDY_TEAM = 2, 0, 1, 2, 3
GREEDY_WAVE_SHARED, GREEDY_WAVE_PLAIN = 4, 5
OPT_PREFIX_SORT = 3
OPT_GREEDY_LDS_PAD = 5
OPT_CONTENT_CHOICE, CONTENT_TARGET, CONTENT_JOINT, CONTENT_JOINT_SPLIT = 6, 0, 1, 2
OPT_FOURTH_OFFSET = 7
OPT_STREAM_ORDER = 8

# every symbol include/iivision.h declares
SYMBOLS = [
    "iiv_version", "iiv_last_error", "iiv_device_count",
    "iiv_masked_bits", "iiv_masked_dots", "iiv_num_offsets", "iiv_table_entries",
    "iiv_store_table_entries",
    "iiv_cie2000_matrix", "iiv_delta_e_cie2000", "iiv_pixel_strings", "iiv_build_table", "iiv_build_store_table",
    "iiv_symmetrise_table", "iiv_store_table_from_table",
    "iiv_pack", "iiv_diff_weights", "iiv_compute_delta_pages",
    "iiv_encoder_create", "iiv_encoder_destroy", "iiv_encoder_set_option", "iiv_encoder_info",
    "iiv_encoder_snapshot", "iiv_encoder_rollback", "iiv_encoder_snapshot_slot", "iiv_encoder_rollback_slot", "iiv_encoder_get_state", "iiv_encoder_set_state",
    "iiv_encoder_set_state_range", "iiv_encoder_get_video_state", "iiv_encoder_set_video_state",
    "iiv_encoder_get_video_brief", "iiv_encoder_get_video_brief_async",
    "iiv_encode", "iiv_encode_streams", "iiv_encoder_live_queue", "iiv_encode_live", "iiv_encoder_set_state_async",
    "iiv_encoder_check", "iiv_encoder_profile", "iiv_encoder_profile_read", "iiv_encoder_input_stats",
    "iiv_encoder_launch_forms",
    "iiv_build_split_store_table", "iiv_split_table_entries", "iiv_check_split_diff_table",
    "iiv_build_narrow_store_table",
    "iiv_check_diff_weight_pieces",
    "iiv_emit_stream", "iiv_emit_chunk", "iiv_frames_to_memory_maps",
]


class Segment(C.Structure):
    _fields_ = [("frame", C.c_int32), ("is_aux", C.c_int32), ("restart", C.c_int32), ("n_ops", C.c_int32)]


class VideoState(C.Structure):
    """include/iivision.h: iiv_video_state"""
    _fields_ = [("mem_main", C.c_uint8 * 8192), ("mem_aux", C.c_uint8 * 8192),
                ("up_main", C.c_int32 * 8192), ("up_aux", C.c_int32 * 8192),
                ("rng_py", C.c_uint32 * 625), ("rng_np", C.c_uint32 * 625),
                ("out_of_work", C.c_int32 * 2), ("packed", C.c_uint64 * 4096)]

    def array(self, name, dtype, shape):
        return np.frombuffer(getattr(self, name), dtype=dtype).reshape(shape)


class VideoBrief(C.Structure):
    """include/iivision.h: iiv_video_brief"""
    _fields_ = [("priority_sum", C.c_int64 * 2), ("hole_bytes", C.c_int32 * 2), ("out_of_work", C.c_int32 * 2),
                ("rng_py", C.c_uint32 * 625), ("rng_np", C.c_uint32 * 625)]


class IIVError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libiivision error %d: %s" % (code, msg))
        self.code = code


class IIVAssertionError(AssertionError):
    """One of the reference's `assert`s fired on the device."""


_lib = None


def lib():
    """Load libiivision.so; fails loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s not found: build it with `make -C ii-vision_amd/csrc` "
            "(or python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback." % LIB_PATH)
    # torch bundles its own HIP runtime (SONAME libamdhip64.so.7).  Import it first so
    # that our NEEDED entry resolves to that already-loaded copy: two HIP/HSA runtimes
    # in one process cannot both own the device.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, i32, sz = C.c_void_p, C.c_int, C.c_size_t
    L.iiv_version.restype = C.c_char_p
    L.iiv_last_error.restype = C.c_char_p
    L.iiv_device_count.restype = i32
    for f in ("iiv_masked_bits", "iiv_masked_dots", "iiv_num_offsets"):
        getattr(L, f).restype = i32
        getattr(L, f).argtypes = [i32]
    for f in ("iiv_table_entries", "iiv_store_table_entries"):
        getattr(L, f).restype = sz
        getattr(L, f).argtypes = [i32]
    L.iiv_cie2000_matrix.argtypes = [vp, vp, vp, vp]
    L.iiv_delta_e_cie2000.argtypes = [i32, vp, vp, vp, vp]
    L.iiv_pixel_strings.argtypes = [i32, vp, vp, vp]
    L.iiv_build_table.argtypes = [i32, vp, vp, i32, vp]
    L.iiv_build_store_table.argtypes = [i32, vp, vp, vp]
    L.iiv_symmetrise_table.argtypes = [i32, vp, vp]
    L.iiv_store_table_from_table.argtypes = [i32, vp, vp, vp]
    L.iiv_pack.argtypes = [i32, i32, vp, vp, vp, vp]
    L.iiv_diff_weights.argtypes = [i32, vp, i32, vp, vp, i32, vp, vp]
    L.iiv_compute_delta_pages.argtypes = [i32, vp, i32, vp, vp, vp, vp, i32, vp, vp]
    L.iiv_encoder_create.argtypes = [i32, vp, vp, vp, i32, C.POINTER(vp)]
    L.iiv_encoder_set_option.argtypes = [vp, i32, i32]
    if hasattr(L, "iiv_encoder_info") or "IIV_LIB" not in os.environ:
        L.iiv_encoder_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.iiv_encoder_snapshot.argtypes = [vp, vp]
    L.iiv_encoder_rollback.argtypes = [vp, vp]
    L.iiv_encoder_snapshot_slot.argtypes = [vp, i32, vp]
    L.iiv_encoder_rollback_slot.argtypes = [vp, i32, vp]
    L.iiv_encoder_destroy.argtypes = [vp]
    L.iiv_encoder_destroy.restype = None
    L.iiv_encoder_get_state.argtypes = [vp, i32, i32, vp, sz]
    L.iiv_encoder_set_state.argtypes = [vp, i32, i32, vp, sz]
    L.iiv_encoder_set_state_range.argtypes = [vp, i32, i32, i32, vp, sz]
    L.iiv_encoder_get_video_state.argtypes = [vp, i32, C.POINTER(VideoState)]
    L.iiv_encoder_get_video_brief.argtypes = [vp, i32, C.POINTER(VideoBrief)]
    if hasattr(L, "iiv_encoder_get_video_brief_async") or "IIV_LIB" not in os.environ:
        L.iiv_encoder_get_video_brief_async.argtypes = [vp, i32, C.POINTER(VideoBrief), vp]
    L.iiv_encoder_set_video_state.argtypes = [vp, i32, C.POINTER(VideoState)]
    L.iiv_encode.argtypes = [vp, vp, vp, i32, C.POINTER(Segment), i32, vp, vp]
    L.iiv_encode_streams.argtypes = [vp, vp, vp, i32, C.POINTER(Segment), C.POINTER(C.c_int32), vp, sz, vp]
    if hasattr(L, "iiv_encode_live") or "IIV_LIB" not in os.environ:
        L.iiv_encoder_live_queue.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(C.c_int)]
        L.iiv_encode_live.argtypes = [vp, vp, vp, i32, C.POINTER(Segment), i32, vp, i32, C.c_uint32, vp]
    if hasattr(L, "iiv_encoder_set_state_async") or "IIV_LIB" not in os.environ:
        L.iiv_encoder_set_state_async.argtypes = [vp, i32, i32, vp, sz, vp]
    L.iiv_build_split_store_table.argtypes = [i32, vp, vp, vp, vp, vp]
    L.iiv_split_table_entries.restype = sz
    L.iiv_check_split_diff_table.argtypes = [i32, vp, vp, vp, vp]
    L.iiv_build_narrow_store_table.argtypes = [i32, vp, vp, vp, vp, vp]
    if hasattr(L, "iiv_check_diff_weight_pieces") or "IIV_LIB" not in os.environ:   # (IIV_LIB: A/B runs against older builds, tools/ab_libs.sh)
        L.iiv_check_diff_weight_pieces.argtypes = [i32, vp, vp, vp, vp]
    L.iiv_split_table_entries.argtypes = [i32, i32]
    L.iiv_encoder_check.argtypes = [vp, C.POINTER(i32), vp]
    L.iiv_encoder_profile.argtypes = [vp, i32]
    L.iiv_encoder_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.iiv_encoder_input_stats.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.iiv_emit_stream.argtypes = [i32, i32, C.c_long, vp, vp, vp, C.c_uint16, C.c_uint16, C.c_long, vp, sz,
                                  C.POINTER(sz), vp]
    L.iiv_emit_chunk.argtypes = [i32, i32, C.c_long, C.c_long, vp, sz, vp, sz, i32, vp, C.c_uint16, vp, sz,
                                 C.POINTER(sz), C.POINTER(sz), vp, vp]
    L.iiv_frames_to_memory_maps.argtypes = [i32, vp, i32, vp, i32, vp, vp, vp]
    if hasattr(L, "iiv_encoder_launch_forms") or "IIV_LIB" not in os.environ:
        L.iiv_encoder_launch_forms.argtypes = [vp, C.POINTER(C.c_int64)]
    for name in SYMBOLS:
        if "IIV_LIB" in os.environ and name in ("iiv_encoder_launch_forms", "iiv_check_diff_weight_pieces", "iiv_encoder_info", "iiv_encoder_get_video_brief_async",
                                                 "iiv_encoder_live_queue", "iiv_encode_live", "iiv_encoder_set_state_async") and not hasattr(L, name):
            continue   # (an older build under IIV_LIB: tools/ab_libs.sh)
        getattr(L, name)  # AttributeError if the library lacks a declared symbol
    _lib = L
    return L


def build_id():
    """The library's build id (include/iivision.h: iiv_version): a hash of the sources and flags it was built from."""
    v = lib().iiv_version().decode("ascii", "replace")
    return v.split(" build ", 1)[1] if " build " in v else "unknown"


def check(rc):
    if rc == OK:
        return
    msg = lib().iiv_last_error().decode("utf-8", "replace")
    if rc == ERR_ASSERT:
        raise IIVAssertionError(msg)
    raise IIVError(rc, msg)


_torch_mod = None


def _torch():
    global _torch_mod
    if _torch_mod is not None:   # (availability was established once; the check costs microseconds per call)
        return _torch_mod
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("libiivision needs an AMD GPU (torch.cuda.is_available() is False); "
                           "there is no CPU fallback")
    _torch_mod = torch
    return torch


def stream_ptr():
    """torch's current stream on the current device, as a raw hipStream_t"""
    torch = _torch()
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:   # the same value as current_stream().cuda_stream without building a Stream object
        return C.c_void_p(raw(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def hptr(a):
    return a.ctypes.data_as(C.c_void_p)


# ---- P1 -------------------------------------------------------------------------

def cie2000_matrix(rgb):
    """rgb: (16,3) uint8 indexed by HGRColours value -> (float64 (16,16), int32 (16,16))."""
    _torch()
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8).reshape(48)
    f = np.zeros((16, 16), dtype=np.float64)
    i = np.zeros((16, 16), dtype=np.int32)
    check(lib().iiv_cie2000_matrix(hptr(rgb), hptr(f), hptr(i), stream_ptr()))
    return f, i


def delta_e_cie2000(lab1, lab2):
    """Delta-E 2000 of (n, 3) Lab pairs on the device (the table builder's own function)."""
    _torch()
    a = np.ascontiguousarray(lab1, dtype=np.float64).reshape(-1, 3)
    b = np.ascontiguousarray(lab2, dtype=np.float64).reshape(-1, 3)
    out = np.zeros(len(a), dtype=np.float64)
    check(lib().iiv_delta_e_cie2000(len(a), hptr(a), hptr(b), hptr(out), stream_ptr()))
    return out


def pixel_strings(mode):
    torch = _torch()
    L = lib()
    n = L.iiv_num_offsets(mode) << L.iiv_masked_bits(mode)
    nd = L.iiv_masked_dots(mode)
    dots = torch.empty(n, dtype=torch.int32, device="cuda")
    pix = torch.empty((n, nd), dtype=torch.uint8, device="cuda")
    check(L.iiv_pixel_strings(mode, dptr(dots), dptr(pix), stream_ptr()))
    shape = (L.iiv_num_offsets(mode), 1 << L.iiv_masked_bits(mode))
    return dots.view(shape), pix.view(shape + (nd,))


def build_table(mode, dm, symmetric=True):
    """Device tensor (num_offsets, 2**(2*bits)) int16-typed storage of the u16 table."""
    torch = _torch()
    L = lib()
    dm = np.ascontiguousarray(dm, dtype=np.int32).reshape(256)
    bits = L.iiv_masked_bits(mode)
    out = torch.empty((L.iiv_num_offsets(mode), 1 << (2 * bits)), dtype=torch.int16, device="cuda")
    check(L.iiv_build_table(mode, hptr(dm), dptr(out), 1 if symmetric else 0, stream_ptr()))
    return out


def build_store_table(mode, dm):
    torch = _torch()
    L = lib()
    dm = np.ascontiguousarray(dm, dtype=np.int32).reshape(256)
    out = torch.empty(L.iiv_store_table_entries(mode), dtype=torch.int16, device="cuda")
    check(L.iiv_build_store_table(mode, hptr(dm), dptr(out), stream_ptr()))
    return out


def check_split_diff_table(mode, dm, table):
    """Number of entries of the full symmetric table that differ from the combination of the
    two halves of the split diff-weight table built from dm (0 = exact everywhere)."""
    _torch()
    dm = np.ascontiguousarray(dm, dtype=np.int32).reshape(256)
    n = C.c_ulonglong(0)
    check(lib().iiv_check_split_diff_table(mode, hptr(dm), dptr(table), C.byref(n), stream_ptr()))
    return int(n.value)


def check_diff_weight_pieces(mode, dm, table):
    """Entries of the full symmetric table that differ from the sum of pixel-pair terms the prologue
    evaluates instead of the recurrence (include/iivision.h: iiv_check_diff_weight_pieces); both modes."""
    dm = np.ascontiguousarray(dm, dtype=np.int32).reshape(256)
    n = C.c_ulonglong(0)
    check(lib().iiv_check_diff_weight_pieces(mode, hptr(dm), dptr(table), C.byref(n), stream_ptr()))
    return int(n.value)


def build_narrow_store_table(mode, dm, store):
    """(expanded, n_mismatch): every store value as the greedy kernels obtain it from the narrow
    form of the split table (S = L1 + RF, two u16 tables), and the number of entries that differ
    from `store` (0 when both come from the same dm)."""
    torch = _torch()
    dm = np.ascontiguousarray(dm, dtype=np.int32).reshape(256)
    exp = torch.empty_like(store)
    n = C.c_ulonglong(0)
    check(lib().iiv_build_narrow_store_table(mode, hptr(dm), dptr(store), dptr(exp), C.byref(n), stream_ptr()))
    return exp, int(n.value)


def build_split_store_table(mode, dm, expanded=True):
    """(left, right, expanded) device tensors: the two halves of the split store table and
    (optionally) the dense store table rebuilt from them with the encoder's index arithmetic."""
    torch = _torch()
    L = lib()
    dm = np.ascontiguousarray(dm, dtype=np.int32).reshape(256)
    left = torch.empty(L.iiv_split_table_entries(mode, 0), dtype=torch.int32, device="cuda")
    right = torch.empty(L.iiv_split_table_entries(mode, 1), dtype=torch.int32, device="cuda")
    exp = torch.empty(L.iiv_store_table_entries(mode), dtype=torch.int16, device="cuda") if expanded else None
    check(L.iiv_build_split_store_table(mode, hptr(dm), dptr(left), dptr(right), dptr(exp), stream_ptr()))
    return left, right, exp


def load_table(mode, lower_or_file_array):
    """A table as a reference .npz holds it (lower triangle) -> (symmetric table, store table) in HBM:
    Bitmap.edit_distances' load + mirror (screen.py:343-367) on the device."""
    torch = _torch()
    L = lib()
    bits = L.iiv_masked_bits(mode)
    a = np.ascontiguousarray(lower_or_file_array, dtype=np.uint16)
    if a.shape != (L.iiv_num_offsets(mode), 1 << (2 * bits)):
        raise ValueError("edit_distance array has shape %s, expected %s" % (a.shape, (L.iiv_num_offsets(mode), 1 << (2 * bits))))
    table = torch.from_numpy(a.view(np.int16)).cuda()
    check(L.iiv_symmetrise_table(mode, dptr(table), stream_ptr()))
    store = torch.empty(L.iiv_store_table_entries(mode), dtype=torch.int16, device="cuda")
    check(L.iiv_store_table_from_table(mode, dptr(table), dptr(store), stream_ptr()))
    return table, store


def table_to_numpy(t):
    """u16 view of a table tensor on the host."""
    return t.cpu().numpy().view(np.uint16)


# ---- P2 -------------------------------------------------------------------------

def pack(mode, main_mem, aux_mem=None):
    """main/aux: uint8 arrays (..., 32, 256) on host -> uint64 (..., 32, 128) on host."""
    torch = _torch()
    main_mem = np.ascontiguousarray(main_mem, dtype=np.uint8)
    lead = main_mem.shape[:-2]
    n = int(np.prod(lead)) if lead else 1
    dm_ = torch.from_numpy(main_mem.reshape(n, 32, 256)).cuda()
    da = None
    if mode == DHGR:
        da = torch.from_numpy(np.ascontiguousarray(aux_mem, dtype=np.uint8).reshape(n, 32, 256)).cuda()
    out = torch.empty((n, 32, 128), dtype=torch.int64, device="cuda")
    check(lib().iiv_pack(mode, n, dptr(dm_), dptr(da), dptr(out), stream_ptr()))
    return out.cpu().numpy().view(np.uint64).reshape(lead + (32, 128))


def diff_weights(mode, table, src_packed, tgt_packed, is_aux):
    torch = _torch()
    src = torch.from_numpy(np.ascontiguousarray(src_packed, dtype=np.uint64).view(np.int64).reshape(1, 32, 128)).cuda()
    tgt = torch.from_numpy(np.ascontiguousarray(tgt_packed, dtype=np.uint64).view(np.int64).reshape(1, 32, 128)).cuda()
    out = torch.empty((32, 256), dtype=torch.int32, device="cuda")
    check(lib().iiv_diff_weights(mode, dptr(table), 1, dptr(src), dptr(tgt), int(bool(is_aux)), dptr(out),
                                 stream_ptr()))
    return out.cpu().numpy()


def compute_delta_pages(mode, table, tgt_packed, pages, contents, dw_rows, is_aux):
    torch = _torch()
    tgt = torch.from_numpy(np.ascontiguousarray(tgt_packed, dtype=np.uint64).view(np.int64)).cuda()
    pages = np.ascontiguousarray(pages, dtype=np.int32).reshape(-1)
    n = len(pages)
    dp = torch.from_numpy(pages).cuda()
    dc = torch.from_numpy(np.ascontiguousarray(contents, dtype=np.int32).reshape(-1)).cuda()
    dr = torch.from_numpy(np.ascontiguousarray(dw_rows, dtype=np.int32).reshape(n, 256)).cuda()
    out = torch.empty((n, 256), dtype=torch.int32, device="cuda")
    check(lib().iiv_compute_delta_pages(mode, dptr(table), n, dptr(tgt), dptr(dp), dptr(dc), dptr(dr),
                                        int(bool(is_aux)), dptr(out), stream_ptr()))
    return out.cpu().numpy()


# ---- P3 -------------------------------------------------------------------------

def encoder_info(handle):
    """(mode, n_streams) of the encoder behind an iiv_encoder* (iiv_encoder_info): what its callers' buffers must be sized for."""
    mode, n = C.c_int(-1), C.c_int(0)
    check(lib().iiv_encoder_info(handle if isinstance(handle, C.c_void_p) else C.c_void_p(int(handle)), C.byref(mode), C.byref(n)))
    return mode.value, n.value


def validate_frames(handle, frames_main, frames_aux):
    """iiv_encode / iiv_encode_streams read n_streams x n_frames x 8192 bytes of every bank whatever the caller's tensors
    hold: refuse anything else here instead of letting a kernel read out of bounds.  Returns (n_streams, n_frames)."""
    torch = _torch()
    mode, n_streams = encoder_info(handle)

    def one(t, name):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous()):
            raise ValueError("%s must be a contiguous CUDA uint8 tensor (n_streams, n_frames, 32, 256)" % name)
        if t.dim() != 4 or tuple(t.shape[2:]) != (32, 256):
            raise ValueError("%s has shape %s, not (n_streams, n_frames, 32, 256)" % (name, tuple(t.shape)))
        if int(t.shape[0]) != n_streams:
            raise ValueError("%s holds %d streams, the encoder was created for %d" % (name, int(t.shape[0]), n_streams))

    one(frames_main, "frames_main")
    n_frames = int(frames_main.shape[1])
    if n_frames <= 0:
        raise ValueError("frames_main holds no frame")
    if mode == DHGR:
        if frames_aux is None:
            raise ValueError("DHGR needs frames_aux")
        one(frames_aux, "frames_aux")
        if int(frames_aux.shape[1]) != n_frames:
            raise ValueError("frames_aux holds %d frames per stream, frames_main %d" % (int(frames_aux.shape[1]), n_frames))
    elif frames_aux is not None:
        one(frames_aux, "frames_aux")   # (HGR ignores it; a tensor of the wrong kind is still a caller's mistake)
    return n_streams, n_frames


def validate_ops_out(ops_out, need_bytes):
    torch = _torch()
    if not (isinstance(ops_out, torch.Tensor) and ops_out.is_cuda and ops_out.dtype == torch.uint8 and ops_out.is_contiguous()):
        raise ValueError("ops_out must be a contiguous CUDA uint8 tensor")
    if ops_out.numel() < need_bytes:
        raise ValueError("ops_out holds %d bytes, this call writes %d (n_streams * total opcodes * 6)" % (ops_out.numel(), need_bytes))


class Encoder:
    """n_streams independent video.Video states resident on the GPU."""

    def __init__(self, mode, table, store_table, n_streams=1, dm=None):
        """dm: the 16x16 int diff matrix the tables came from; if given, diff weights
        are recomputed by recurrence instead of gathered from `table` (same values)."""
        _torch()
        self.mode = mode
        self.n_streams = int(n_streams)
        self._table = table            # keep the device tensors alive
        self._store = store_table
        h = C.c_void_p()
        dmp = C.c_void_p(0)
        if dm is not None:
            self._dm = np.ascontiguousarray(dm, dtype=np.int32).reshape(256)
            dmp = hptr(self._dm)
        check(lib().iiv_encoder_create(mode, dptr(table), dptr(store_table), dmp, self.n_streams, C.byref(h)))
        self._h = h

    @property
    def handle(self):
        """The iiv_encoder* as an int: what torch.ops.iivision.encode / encode_streams take (torch_ops.py)."""
        return int(self._h.value)

    def set_greedy_kernel(self, wave_per_stream):
        """True: one wave per stream; False: one 256-thread workgroup; "team": eight waves per stream;
        "shared" / "plain": one wave per stream with / without the bank's L1 table half shared in LDS by
        the eight streams of a workgroup (True picks by batch size); None: automatic."""
        names = {None: GREEDY_AUTO, "auto": GREEDY_AUTO, "team": GREEDY_TEAM, "shared": GREEDY_WAVE_SHARED, "plain": GREEDY_WAVE_PLAIN,
                 "wave": GREEDY_WAVE, "workgroup": GREEDY_WORKGROUP}
        if wave_per_stream is None or isinstance(wave_per_stream, str):
            if wave_per_stream not in names:
                raise ValueError("set_greedy_kernel: unknown kernel %r (one of %s, True, False, None)"
                                 % (wave_per_stream, ", ".join(repr(k) for k in names if k)))
            v = names[wave_per_stream]
        else:
            v = GREEDY_WAVE if wave_per_stream else GREEDY_WORKGROUP
        check(lib().iiv_encoder_set_option(self._h, OPT_GREEDY_KERNEL, v))

    def set_stream_order(self, enable):
        """True (default): big batches launch the one-wave kernel longest stream first (IIV_OPT_STREAM_ORDER); same bytes."""
        check(lib().iiv_encoder_set_option(self._h, OPT_STREAM_ORDER, 1 if enable else 0))

    def set_prefix_sort(self, enable):
        check(lib().iiv_encoder_set_option(self._h, OPT_PREFIX_SORT, 1 if enable else 0))

    def set_greedy_lds_pad(self, n_bytes):
        check(lib().iiv_encoder_set_option(self._h, OPT_GREEDY_LDS_PAD, int(n_bytes)))

    def set_content_choice(self, joint):
        """False (default): the reference's greedy step.  True: the content byte of every step is
        chosen jointly with its extra offsets (include/iivision.h: IIV_CONTENT_JOINT) -- the
        reference README's "global optimization" idea, NOT the reference's output.  "split": the same choice
        computed by the slower second implementation (IIV_CONTENT_JOINT_SPLIT)."""
        value = CONTENT_JOINT_SPLIT if joint == "split" else CONTENT_JOINT if joint else CONTENT_TARGET
        check(lib().iiv_encoder_set_option(self._h, OPT_CONTENT_CHOICE, value))

    def set_fourth_offset(self, enable):
        """f4: up to three extra offsets per opcode instead of two and a copy of the first (the reference's exit test
        `len(offsets) == 3`, video.py:180-181, read as 4; include/iivision.h IIV_OPT_FOURTH_OFFSET).  NOT the
        reference's opcode stream."""
        check(lib().iiv_encoder_set_option(self._h, OPT_FOURTH_OFFSET, 1 if enable else 0))

    def set_diff_weights_mode(self, mode):
        """True / "recurrence" (default with dm): the edit-distance recurrence in the kernel;
        "split": two gathers from the split diff-weight table; False / "table": one gather from
        the full table."""
        v = {"split": DW_SPLIT, "recurrence": DW_RECURRENCE, "table": DW_TABLE, True: DW_RECURRENCE,
             False: DW_TABLE}[mode]
        check(lib().iiv_encoder_set_option(self._h, OPT_DIFF_WEIGHTS, v))

    def close(self):
        if getattr(self, "_h", None):
            lib().iiv_encoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    _ITEMS = {
        STATE_MEM_MAIN: ((32, 256), np.uint8), STATE_MEM_AUX: ((32, 256), np.uint8),
        STATE_UP_MAIN: ((32, 256), np.int32), STATE_UP_AUX: ((32, 256), np.int32),
        STATE_RNG_PY: ((625,), np.uint32), STATE_RNG_NP: ((625,), np.uint32),
        STATE_OUT_OF_WORK: ((2,), np.int32), STATE_PACKED: ((32, 128), np.uint64),
        STATE_COUNTERS: ((4,), np.uint64),
        100: ((32,), np.uint64),  # phase stamps of diagnostic (-DIIV_STAMPS) builds
    }

    def get_state(self, what, stream=0, out=None):
        shape, dt = self._ITEMS[what]
        if out is None:
            out = np.empty(shape, dtype=dt)
        assert out.flags.c_contiguous and out.dtype == dt and out.shape == shape
        check(lib().iiv_encoder_get_state(self._h, int(stream), what, hptr(out), out.nbytes))
        return out

    def set_state(self, what, value, stream=0):
        shape, dt = self._ITEMS[what]
        a = np.ascontiguousarray(value, dtype=dt).reshape(shape)
        check(lib().iiv_encoder_set_state(self._h, int(stream), what, hptr(a), a.nbytes))

    def set_state_async(self, what, value, stream=0):
        """STATE_OUT_OF_WORK / STATE_RNG_PY / STATE_RNG_NP of one stream, enqueued behind the launches already on the current
        HIP stream: no device-wide synchronisation (iiv_encoder_set_state_async)."""
        shape, dt = self._ITEMS[what]
        a = np.ascontiguousarray(value, dtype=dt).reshape(shape)
        check(lib().iiv_encoder_set_state_async(self._h, int(stream), what, hptr(a), a.nbytes, stream_ptr()))

    def get_video_state(self, stream=0, out=None):
        out = out if out is not None else VideoState()
        check(lib().iiv_encoder_get_video_state(self._h, int(stream), C.byref(out)))
        return out

    def get_video_brief(self, stream=0, out=None):
        out = out if out is not None else VideoBrief()
        check(lib().iiv_encoder_get_video_brief(self._h, int(stream), C.byref(out)))
        return out

    def get_video_brief_async(self, out, stream=0):
        """The brief of the state as it will be behind the launches already enqueued, copied into `out` (a VideoBrief in
        pinned memory) by the time the stream is synchronised (check())."""
        check(lib().iiv_encoder_get_video_brief_async(self._h, int(stream), C.byref(out), stream_ptr()))

    def set_video_state(self, state, stream=0):
        check(lib().iiv_encoder_set_video_state(self._h, int(stream), C.byref(state)))

    def set_state_all(self, what, values, first=0):
        """values: (n, ...) array, item `what` of streams first .. first + n - 1 in one upload."""
        shape, dt = self._ITEMS[what]
        a = np.ascontiguousarray(values, dtype=dt)
        n = a.shape[0]
        a = a.reshape((n,) + shape)
        check(lib().iiv_encoder_set_state_range(self._h, int(first), int(n), what, hptr(a), a[0].nbytes))

    def encode_streams(self, frames_main, frames_aux, schedules, ops_out=None):
        """Per-stream schedules: schedules[s] = list of (frame, is_aux, restart, n_ops) of stream s.
        Returns (ops tensor (n_streams, max total, 6), per-stream totals).  Asynchronous."""
        torch = _torch()
        _, n_frames = validate_frames(self._h, frames_main, frames_aux)
        if len(schedules) != self.n_streams:
            raise ValueError("%d schedules for %d streams" % (len(schedules), self.n_streams))
        flat = [g for sch in schedules for g in sch]
        segs = (Segment * max(len(flat), 1))(*[Segment(int(f), int(a), int(r), int(k)) for (f, a, r, k) in flat])
        begin = np.zeros(self.n_streams + 1, dtype=np.int32)
        begin[1:] = np.cumsum([len(sch) for sch in schedules])
        totals = [sum(int(g[3]) for g in sch) for sch in schedules]
        width = max(max(totals), 1)
        if ops_out is None:
            ops_out = torch.zeros((self.n_streams, width, 6), dtype=torch.uint8, device="cuda")
        if ops_out.dim() != 3 or int(ops_out.shape[0]) != self.n_streams or int(ops_out.shape[2]) != 6 or int(ops_out.shape[1]) < width:
            raise ValueError("ops_out has shape %s, this call needs (%d, >= %d, 6)" % (tuple(ops_out.shape), self.n_streams, width))
        validate_ops_out(ops_out, self.n_streams * width * 6)
        stride = ops_out.shape[1] * 6
        check(lib().iiv_encode_streams(self._h, dptr(frames_main), dptr(frames_aux), int(n_frames), segs,
                                       begin.ctypes.data_as(C.POINTER(C.c_int32)), dptr(ops_out), stride, stream_ptr()))
        return ops_out, totals

    def encode(self, frames_main, frames_aux, segments, ops_out=None):
        """frames_*: CUDA uint8 tensors (n_streams, n_frames, 32, 256); segments: list of
        (frame, is_aux, restart, n_ops).  Returns the CUDA uint8 tensor
        (n_streams, total_ops, 6).  Asynchronous."""
        torch = _torch()
        _, n_frames = validate_frames(self._h, frames_main, frames_aux)
        segs = (Segment * max(len(segments), 1))(*[Segment(int(f), int(a), int(r), int(k)) for (f, a, r, k) in segments])
        total = sum(int(s[3]) for s in segments)
        need = self.n_streams * total * 6
        if ops_out is None:
            ops_out = torch.empty((self.n_streams, total, 6), dtype=torch.uint8, device="cuda")
        else:
            # iiv_encode packs stream s at byte s * total * 6 whatever the shape of the caller's buffer: a
            # pre-allocated buffer may be larger than this call needs (calls of a Movie-paced driver differ
            # in their opcode count), never smaller -- that would be a device write out of bounds
            validate_ops_out(ops_out, need)
        check(lib().iiv_encode(self._h, dptr(frames_main), dptr(frames_aux), int(n_frames), segs, len(segments),
                               dptr(ops_out), stream_ptr()))
        # the rows as they were written: (n_streams, total, 6) over the front of the buffer
        return ops_out.view(-1)[:need].view(self.n_streams, total, 6)

    def live_queue(self, slot=0):
        """Queue `slot` of the live hand-over (include/iivision.h: iiv_encoder_live_queue) as a numpy uint64 view of the
        coherent host memory the team kernel writes its opcodes into: slot j = six opcode bytes | tag << 48."""
        addr, cap = C.c_void_p(0), C.c_int(0)
        check(lib().iiv_encoder_live_queue(self._h, int(slot), C.byref(addr), C.byref(cap)))
        return np.frombuffer((C.c_uint64 * cap.value).from_address(addr.value), dtype=np.uint64)

    def encode_live(self, frames_main, frames_aux, segment, ops_out, slot, tag):
        """iiv_encode_live of ONE segment (frame, is_aux, restart, n_ops): iiv_encode, the opcodes also appearing one by one
        in live_queue(slot) under `tag`.  Raises IIVError(ERR_INVALID) -- nothing launched -- if this encoder's options
        keep it off the team kernel."""
        _, n_frames = validate_frames(self._h, frames_main, frames_aux)
        f, a, r, k = segment
        seg = (Segment * 1)(Segment(int(f), int(a), int(r), int(k)))
        validate_ops_out(ops_out, self.n_streams * int(k) * 6)
        check(lib().iiv_encode_live(self._h, dptr(frames_main), dptr(frames_aux), int(n_frames), seg, 1, dptr(ops_out),
                                    int(slot), int(tag), stream_ptr()))

    def snapshot(self, slot=0):
        check(lib().iiv_encoder_snapshot_slot(self._h, int(slot), stream_ptr()))

    def rollback(self, slot=0):
        check(lib().iiv_encoder_rollback_slot(self._h, int(slot), stream_ptr()))

    def check(self):
        bad = C.c_int(-1)
        check(lib().iiv_encoder_check(self._h, C.byref(bad), stream_ptr()))

    def profile(self, enable=True):
        check(lib().iiv_encoder_profile(self._h, 1 if enable else 0))

    def profile_read(self):
        ms = (C.c_double * 2)()
        n = (C.c_int64 * 2)()
        check(lib().iiv_encoder_profile_read(self._h, ms, n))
        return {"prologue_ms": ms[0], "greedy_ms": ms[1], "prologue_launches": n[0], "greedy_launches": n[1]}

    def launch_forms(self):
        """Greedy launches since profile(True) by kernel: {"plain", "shared", "team", "workgroup"} (iiv_encoder_launch_forms)."""
        if not hasattr(lib(), "iiv_encoder_launch_forms"):   # (an older build under IIV_LIB, tools/ab_libs.sh)
            return None
        c = (C.c_int64 * 4)()
        check(lib().iiv_encoder_launch_forms(self._h, c))
        return {"plain": int(c[0]), "shared": int(c[1]), "team": int(c[2]), "workgroup": int(c[3])}

    def input_stats(self):
        """(share of the steps the nonces decided, as the kernels reported it for an earlier call; the form of the one-wave
        kernel the next full-batch launch runs: "shared" / "plain") -- include/iivision.h: iiv_encoder_input_stats"""
        st, form = (C.c_double * 2)(), C.c_int(0)
        check(lib().iiv_encoder_input_stats(self._h, st, C.byref(form)))
        self.real_opcodes_per_launch = st[1]
        return st[0], "shared" if form.value == GREEDY_WAVE_SHARED else "plain"


# ---- f2: byte emission -------------------------------------------------------------

def emit_stream_size(mode, n_ops, tick_addr, ack_addr, terminate_addr, max_bytes_out=None):
    ta = np.ascontiguousarray(tick_addr, dtype=np.uint16).reshape(1024)
    n = C.c_size_t(0)
    check(lib().iiv_emit_stream(mode, 1, int(n_ops), None, None, hptr(ta), int(ack_addr), int(terminate_addr),
                                int(max_bytes_out or 0), None, 0, C.byref(n), None))
    return int(n.value)


def emit_stream(mode, ops, ticks, tick_addr, ack_addr, terminate_addr, max_bytes_out=None):
    """ops: CUDA uint8 (S, n, 6); ticks: CUDA uint8 (S, n) -> CUDA uint8 (S, length)."""
    torch = _torch()
    S, n = int(ops.shape[0]), int(ops.shape[1])
    ta = np.ascontiguousarray(tick_addr, dtype=np.uint16).reshape(1024)
    length = emit_stream_size(mode, n, ta, ack_addr, terminate_addr, max_bytes_out)
    out = torch.empty((S, length), dtype=torch.uint8, device="cuda")
    got = C.c_size_t(0)
    ops_c, ticks_c = ops.contiguous(), ticks.contiguous()   # (kept in locals: the call reads their storage)
    check(lib().iiv_emit_stream(mode, S, n, dptr(ops_c), dptr(ticks_c), hptr(ta), int(ack_addr),
                                int(terminate_addr), int(max_bytes_out or 0), dptr(out), length, C.byref(got),
                                stream_ptr()))
    return out


def emit_chunk_range(mode, first_op, n_ops):
    """(first byte, byte count) of opcodes [first_op, first_op + n_ops) in a stream."""
    b0, nb = C.c_size_t(0), C.c_size_t(0)
    check(lib().iiv_emit_chunk(mode, 1, int(first_op), int(n_ops), None, 0, None, 0, 34, None, 0, None, 0,
                               C.byref(b0), C.byref(nb), None, None))
    return int(b0.value), int(nb.value)


def emit_chunk(mode, ops, first_op, d_tick_addr, ack_addr, out, ticks=None, const_tick=34, d_err=None):
    """Asynchronous slice emission.  ops: CUDA uint8 (S, n, 6) view (row stride arbitrary) holding opcodes
    first_op .. first_op + n - 1 of every stream; out: CUDA uint8 (S, >= byte count) -> (first byte, byte count)."""
    _torch()
    S, n = int(ops.shape[0]), int(ops.shape[1])
    assert ops.stride(2) == 1 and ops.stride(1) == 6 and out.stride(1) == 1
    b0, nb = C.c_size_t(0), C.c_size_t(0)
    check(lib().iiv_emit_chunk(mode, S, int(first_op), n, dptr(ops), int(ops.stride(0)), dptr(ticks),
                               int(ticks.stride(0)) if ticks is not None else 0, int(const_tick), dptr(d_tick_addr),
                               int(ack_addr), dptr(out), int(out.stride(0)), C.byref(b0), C.byref(nb), dptr(d_err),
                               stream_ptr()))
    return int(b0.value), int(nb.value)


# ---- f3: frame ingest ---------------------------------------------------------------

DITHER_DIFFUSION = 256   # IIV_DITHER_DIFFUSION: Floyd-Steinberg error diffusion instead of the ordered dither


def frames_to_memory_maps(mode, palette_rgb, rgb, dither=0, out=None):
    """rgb: CUDA uint8 (n, 192, 280, 3) -> (main, aux) CUDA uint8 (n, 32, 256); aux is None for HGR.
    dither: 0..255 = amplitude of the 4x4 ordered dither, DITHER_DIFFUSION = error diffusion.
    out=(main, aux): write into these contiguous CUDA uint8 tensors of n * 8192 bytes each (any shape; aux ignored
    for HGR) instead of allocating -- e.g. a (streams, frames, 32, 256) slice of a batch's target frames.
    Asynchronous on torch's current stream (since round 5; it used to synchronise): `rgb` and the `out` tensors must stay
    alive and untouched until that stream has reached this call.  Tensors allocated and used on the same current stream are
    safe as they are (the caching allocator re-uses a freed block on that stream only behind its pending work); tensors
    that belong to ANOTHER stream -- a conversion on a side stream writing into the encoder's target buffers, as bench.py's
    e2e leg does -- need tensor.record_stream(torch.cuda.current_stream()) or an event between the streams."""
    torch = _torch()
    if not (rgb.is_cuda and rgb.dtype == torch.uint8 and rgb.dim() == 4 and tuple(rgb.shape[1:]) == (192, 280, 3) and rgb.is_contiguous()):
        raise ValueError("rgb must be a contiguous CUDA uint8 tensor (n, 192, 280, 3)")
    n = int(rgb.shape[0])
    pal = np.ascontiguousarray(palette_rgb, dtype=np.uint8).reshape(48)
    if out is None:
        main = torch.empty((n, 32, 256), dtype=torch.uint8, device="cuda")
        aux = torch.empty((n, 32, 256), dtype=torch.uint8, device="cuda") if mode == DHGR else None
    else:
        main, aux = out
        for t in ((main, aux) if mode == DHGR else (main,)):
            if not (t is not None and t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous() and t.numel() == n * 8192):
                raise ValueError("out tensors must be contiguous CUDA uint8 tensors of n * 8192 bytes")
        if mode != DHGR:
            aux = None
    check(lib().iiv_frames_to_memory_maps(mode, hptr(pal), n, dptr(rgb), int(dither), dptr(main), dptr(aux), stream_ptr()))
    return main, aux
