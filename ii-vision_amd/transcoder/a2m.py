"""Byte emission of the player opcode stream (.a2m), batched on the GPU.

Mirrors what movie.Movie.emit_stream / done (transcoder/movie.py:113-161) produce
through opcodes.py / machine.py for one stream, for any number of streams at once
(iiv_emit_stream, csrc/iiv_a2m.hip).  Audio is out of scope here: the caller supplies
the speaker duty cycle ("tick", 4..66 even, movie.py:104-107) of every opcode.
"""

import numpy as np

import _iiv_native as native
import symbol_table


class OpcodeAddresses:
    """Entry points of the player's opcodes (opcodes.py:168-217)."""

    def __init__(self, tick, ack, terminate, nop=0):
        self.tick = np.ascontiguousarray(tick, dtype=np.uint16).reshape(32, 32)  # [(tick-4)/2][page-32]
        self.ack = int(ack)
        self.terminate = int(terminate)
        self.nop = int(nop)

    @classmethod
    def from_debug_file(cls, path="player/iivision.dbg"):
        """Read `op_*` labels from the cc65 debug file, as opcodes._parse_symbol_table does."""
        syms = symbol_table.SymbolTable(path).parse()
        ops = {}
        for name, data in syms.items():
            if name.startswith('"op_'):
                ops[name[4:-1]] = int(data["val"], 16)
        tick = np.zeros((32, 32), dtype=np.uint16)
        for ti, t in enumerate(range(4, 68, 2)):
            for page in range(32, 64):
                key = "tick_%d_page_%d" % (t, page)
                if key not in ops:
                    raise ValueError("Unable to find opcode address for %s in player debug symbols" % key.upper())
                tick[ti, page - 32] = ops[key]
        for key in ("ack", "terminate", "nop"):
            if key not in ops:
                raise ValueError("Unable to find opcode address for %s in player debug symbols" % key.upper())
        return cls(tick, ops["ack"], ops["terminate"], ops["nop"])


def stream_length(mode, n_ops, addresses, max_bytes_out=None):
    return native.emit_stream_size(mode, n_ops, addresses.tick, addresses.ack, addresses.terminate, max_bytes_out)


def emit_stream(mode, ops, ticks, addresses, max_bytes_out=None):
    """ops: CUDA uint8 (n_streams, n_ops, 6) from Encoder.encode / StreamBatch;
    ticks: CUDA uint8 (n_streams, n_ops) -> CUDA uint8 (n_streams, stream_length)."""
    return native.emit_stream(mode, ops, ticks, addresses.tick, addresses.ack, addresses.terminate, max_bytes_out)
