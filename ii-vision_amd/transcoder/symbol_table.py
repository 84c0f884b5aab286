"""Symbol addresses from a cc65 .dbg file (mirrors transcoder/symbol_table.py).

The player's opcode entry points (`op_tick_<t>_page_<p>`, `op_ack`, `op_terminate`,
...) are data produced by assembling player/main.s; the transcoder only needs their
addresses to emit the byte stream."""

from typing import Dict, TextIO


class SymbolTable:
    """Parse cc65 debug file to extract symbol table."""

    def __init__(self, debugfile: str = None):
        self.debugfile = debugfile  # type: str

    def parse(self, iostream: TextIO = None) -> Dict:
        """name (quoted, as in the file) -> dict of the sym line's key=value fields."""
        stream = iostream if iostream else open(self.debugfile, "r")
        syms = {}
        with stream as f:
            for line in f:
                if not line.startswith("sym"):
                    continue
                fields = dict(kv.split("=", 1) for kv in line.split()[1].split(","))
                syms[fields["name"]] = fields
        return syms
