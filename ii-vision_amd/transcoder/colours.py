"""Nominal Apple II display colours as 4-bit dot windows.

Host-side mirror of transcoder/colours.py (enums :18-71, rotates :74-97, the
sliding-window colour model :100-148).  The table build does not call these: the
same model runs on the GPU for all masked values at once (pixel_kernel in
csrc/iiv_tables.hip); these helpers exist so code written against the reference's
module keeps working, and they are checked against the same golden vectors.
"""

import enum
import functools
from typing import Tuple, Type


class NominalColours(enum.Enum):
    pass


def _members(order):
    names = ("BLACK", "MAGENTA", "BROWN", "ORANGE", "DARK_GREEN", "GREY1", "GREEN", "YELLOW",
             "DARK_BLUE", "VIOLET", "GREY2", "PINK", "MED_BLUE", "LIGHT_BLUE", "AQUA", "WHITE")
    return dict(zip(names, order))


# Dots are in memory bit order (MSB -> LSB); values per colours.py:27-42.
HGRColours = enum.Enum(
    "HGRColours",
    _members((0b0000, 0b0001, 0b1000, 0b1001, 0b0100, 0b0101, 0b1100, 0b1101,
              0b0010, 0b0011, 0b1010, 0b1011, 0b0110, 0b0111, 0b1110, 0b1111)),
    type=NominalColours)

# One tick of colour-reference phase later: HGR values rotated right (colours.py:55-70).
DHGRColours = enum.Enum(
    "DHGRColours",
    _members((0b0000, 0b1000, 0b0100, 0b1100, 0b0010, 0b1010, 0b0110, 0b1110,
              0b0001, 0b1001, 0b0101, 0b1101, 0b0011, 0b1011, 0b0111, 0b1111)),
    type=NominalColours)


def ror(int4: int, howmany: int) -> int:
    """Rotate a 4-bit value right `howmany` times."""
    k = howmany % 4
    return ((int4 >> k) | (int4 << (4 - k))) & 0xf


def rol(int4: int, howmany: int) -> int:
    """Rotate a 4-bit value left `howmany` times."""
    k = howmany % 4
    return ((int4 << k) | (int4 >> (4 - k))) & 0xf


def window_values(num_bits: int, dots: int, init_phase: int = 1) -> Tuple[int, ...]:
    """Colour value of every pixel: the 4-dot window starting at dot k, rotated
    left by the NTSC phase (init_phase + k) mod 4."""
    return tuple(rol((dots >> k) & 0xf, (init_phase + k) & 3) for k in range(num_bits))


@functools.lru_cache(None)
def dots_to_nominal_colour_pixels(
        num_bits: int,
        dots: int,
        colours: Type[NominalColours],
        init_phase: int = 1
) -> Tuple[NominalColours, ...]:
    """num_bits nominal colour pixels via the sliding 4-bit window (colours.py:100-134)."""
    return tuple(colours(v) for v in window_values(num_bits, dots, init_phase))


@functools.lru_cache(None)
def dots_to_nominal_colour_pixel_values(
        num_bits: int,
        dots: int,
        colours: Type[NominalColours],
        init_phase: int = 1
) -> Tuple[int, ...]:
    """Same, as enum values (colours.py:137-148)."""
    return tuple(p.value for p in dots_to_nominal_colour_pixels(num_bits, dots, colours, init_phase))
