"""`torch.ops.iivision.*`: the hot entry points of libiivision.so as PyTorch custom operators.

BASELINE.json's north_star asks for "hand-written HIP kernels via PyTorch-ROCm custom ops"; the contract of the
library stays the C ABI (include/iivision.h) and these operators are a thin registration over it: tensors in,
tensors out / mutated, the launch stream taken from torch's current stream, no kernel of their own.  PyTorch is
plumbing here -- device memory, streams, the dispatcher -- not the product.

    import torch_ops                                   # registers the operators (needs no GPU)
    torch.ops.iivision.encode(handle, frames_main, frames_aux, segments, ops_out)

Operators (schema -> C entry point -> what it replaces in the reference):
    cie2000_matrix(rgb) -> (f64[16,16], i32[16,16])            iiv_cie2000_matrix      make_data_tables.py:55-70
    build_table(mode, dm, symmetric) -> u16-as-i16 tensor      iiv_build_table         make_data_tables.py:111-174 (+ screen.py:358-365)
    build_store_table(mode, dm) -> tensor                      iiv_build_store_table   Bitmap.compute_delta_page's values
    encode(handle, main, aux?, segments, ops_out!) -> ()       iiv_encode              video.py:72-301 under movie.py:56-111
    encode_streams(handle, main, aux?, segments, seg_begin, ops_out!) -> ()   iiv_encode_streams   the same, a schedule per stream
    emit_chunk(mode, ops, first_op, tick_addr, ack_addr, const_tick, out!) -> (first byte, byte count)   iiv_emit_chunk   movie.py:113-161
    frames_to_memory_maps(mode, palette_rgb, rgb, dither) -> (main, aux)      iiv_frames_to_memory_maps   frame_grabber.py:68-115

`handle` is Encoder.handle (the iiv_encoder* as an int): the encoder object -- tables, stream states, options --
is created and owned by _iiv_native.Encoder.  `segments` is an int32 CPU tensor (n, 4) of (frame, is_aux, restart,
n_ops) rows -- the launch plan is host data, exactly as in the C ABI.  There is no CPU implementation: the
operators raise without a GPU.
"""

from typing import Optional, Tuple

import numpy as np
import torch
from torch import Tensor

import _iiv_native as native

NAMES = ("cie2000_matrix", "build_table", "build_store_table", "encode", "encode_streams", "emit_chunk",
         "frames_to_memory_maps")


def _segments_struct(segments: Tensor):
    s = segments.detach().to("cpu", torch.int32).contiguous().view(-1, 4)
    arr = (native.Segment * max(int(s.shape[0]), 1))()
    if s.shape[0]:
        native.C.memmove(arr, s.data_ptr(), int(s.shape[0]) * 16)
    return arr, s


@torch.library.custom_op("iivision::cie2000_matrix", mutates_args=())
def cie2000_matrix(rgb: Tensor) -> Tuple[Tensor, Tensor]:
    f, i = native.cie2000_matrix(rgb.detach().cpu().numpy())
    return torch.from_numpy(f), torch.from_numpy(i)


@torch.library.custom_op("iivision::build_table", mutates_args=())
def build_table(mode: int, dm: Tensor, symmetric: bool) -> Tensor:
    return native.build_table(mode, dm.detach().cpu().numpy(), symmetric)


@torch.library.custom_op("iivision::build_store_table", mutates_args=())
def build_store_table(mode: int, dm: Tensor) -> Tensor:
    return native.build_store_table(mode, dm.detach().cpu().numpy())


@torch.library.custom_op("iivision::encode", mutates_args=("ops_out",), device_types="cuda")
def encode(handle: int, frames_main: Tensor, frames_aux: Optional[Tensor], segments: Tensor, ops_out: Tensor) -> None:
    # (the stream count is the ENCODER's -- iiv_encode reads and writes that many streams whatever the tensors hold)
    n_streams, n_frames = native.validate_frames(handle, frames_main, frames_aux)
    segs, s = _segments_struct(segments)
    total = int(s[:, 3].sum()) if s.shape[0] else 0
    native.validate_ops_out(ops_out, n_streams * total * 6)
    native.check(native.lib().iiv_encode(native.C.c_void_p(handle), native.dptr(frames_main), native.dptr(frames_aux), n_frames,
                                         segs, int(s.shape[0]), native.dptr(ops_out), native.stream_ptr()))


@torch.library.custom_op("iivision::encode_streams", mutates_args=("ops_out",), device_types="cuda")
def encode_streams(handle: int, frames_main: Tensor, frames_aux: Optional[Tensor], segments: Tensor, seg_begin: Tensor,
                   ops_out: Tensor) -> None:
    n_streams, n_frames = native.validate_frames(handle, frames_main, frames_aux)
    segs, s = _segments_struct(segments)
    begin = seg_begin.detach().to("cpu", torch.int32).contiguous().view(-1)
    if int(begin.numel()) != n_streams + 1:
        raise ValueError("seg_begin holds %d entries, the encoder's %d streams need %d" % (int(begin.numel()), n_streams, n_streams + 1))
    b = begin.numpy()
    if int(b[0]) < 0 or (np.diff(b) < 0).any() or int(b[-1]) > int(s.shape[0]):
        raise ValueError("seg_begin must ascend from >= 0 to <= the number of segments (%d)" % int(s.shape[0]))
    if ops_out.dim() != 3 or int(ops_out.shape[0]) != n_streams or int(ops_out.shape[2]) != 6:
        raise ValueError("ops_out has shape %s, not (%d streams, max opcodes, 6)" % (tuple(ops_out.shape), n_streams))
    native.validate_ops_out(ops_out, n_streams * int(ops_out.shape[1]) * 6)
    native.check(native.lib().iiv_encode_streams(native.C.c_void_p(handle), native.dptr(frames_main), native.dptr(frames_aux),
                                                 n_frames, segs,
                                                 native.C.cast(begin.data_ptr(), native.C.POINTER(native.C.c_int32)),
                                                 native.dptr(ops_out), int(ops_out.shape[1]) * 6, native.stream_ptr()))


@torch.library.custom_op("iivision::emit_chunk", mutates_args=("out",), device_types="cuda")
def emit_chunk(mode: int, ops: Tensor, first_op: int, tick_addr: Tensor, ack_addr: int, const_tick: int, out: Tensor) -> Tuple[int, int]:
    return native.emit_chunk(mode, ops, first_op, tick_addr, ack_addr, out, const_tick=const_tick)


@torch.library.custom_op("iivision::frames_to_memory_maps", mutates_args=(), device_types="cuda")
def frames_to_memory_maps(mode: int, palette_rgb: Tensor, rgb: Tensor, dither: int) -> Tuple[Tensor, Tensor]:
    main, aux = native.frames_to_memory_maps(mode, palette_rgb.detach().cpu().numpy(), rgb, dither)
    # (an operator returns tensors: HGR has no aux bank -> an empty one)
    return main, aux if aux is not None else main.new_empty((0, 32, 256))


def encode_via_op(enc, frames_main, frames_aux, segments, ops_out=None):
    """Encoder.encode's contract through torch.ops.iivision.encode: segments as a list of (frame, is_aux, restart, n_ops);
    returns the (n_streams, total, 6) view of the rows as they were written."""
    total = sum(int(s[3]) for s in segments)
    need = enc.n_streams * total * 6
    if int(frames_main.shape[0]) != enc.n_streams:
        raise ValueError("frames_main holds %d streams, the encoder was created for %d" % (int(frames_main.shape[0]), enc.n_streams))
    if ops_out is None:
        ops_out = torch.empty((enc.n_streams, total, 6), dtype=torch.uint8, device="cuda")
    seg_t = torch.tensor([[int(v) for v in s] for s in segments], dtype=torch.int32).view(-1, 4)
    torch.ops.iivision.encode(enc.handle, frames_main, frames_aux, seg_t, ops_out)
    return ops_out.view(-1)[:need].view(enc.n_streams, total, 6)
