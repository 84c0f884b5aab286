"""RGB palettes for rendering NominalColour pixels (mirrors transcoder/palette.py).

The reference stores colormath sRGBColor objects; colormath is not a dependency
here (the CIE2000 arithmetic runs on the GPU, csrc/iiv_tables.hip), so RGB is a
small value class with the attributes the reference's users touch.
"""

import enum
from typing import Dict, Type

import numpy as np

from colours import HGRColours


class RGB:
    """8-bit sRGB triple; rgb_r/g/b are the 0..1 floats colormath would hold."""

    __slots__ = ("r", "g", "b")

    def __init__(self, r, g, b):
        self.r, self.g, self.b = int(r), int(g), int(b)

    rgb_r = property(lambda self: self.r / 255.0)
    rgb_g = property(lambda self: self.g / 255.0)
    rgb_b = property(lambda self: self.b / 255.0)

    def get_upscaled_value_tuple(self):
        return (self.r, self.g, self.b)

    def __repr__(self):
        return "RGB(%d, %d, %d)" % (self.r, self.g, self.b)


def rgb(r, g, b):
    return RGB(r, g, b)


class Palette(enum.Enum):
    """BMP2DHR palette numbers (palette.py:18-23)."""
    UNKNOWN = -1
    IIGS = 0
    NTSC = 5


class BasePalette:
    ID = Palette.UNKNOWN  # type: Palette
    RGB = {}  # type: Dict[HGRColours, RGB]

    @classmethod
    def rgb_array(cls) -> np.ndarray:
        """(16, 3) uint8, row i = colour whose HGRColours value is i."""
        out = np.zeros((16, 3), dtype=np.uint8)
        for colour, v in cls.RGB.items():
            out[colour.value] = v.get_upscaled_value_tuple()
        return out


def _table(rows):
    names = ("BLACK", "MAGENTA", "BROWN", "ORANGE", "DARK_GREEN", "GREY1", "GREEN", "YELLOW",
             "DARK_BLUE", "VIOLET", "GREY2", "PINK", "MED_BLUE", "LIGHT_BLUE", "AQUA", "WHITE")
    return {HGRColours[n]: rgb(*v) for n, v in zip(names, rows)}


class NTSCPalette(BasePalette):
    """BMP2DHGR's default NTSC palette (palette.py:33-54)."""
    ID = Palette.NTSC
    RGB = _table(((0, 0, 0), (148, 12, 125), (99, 77, 0), (249, 86, 29), (51, 111, 0), (126, 126, 126),
                  (67, 200, 0), (221, 206, 23), (32, 54, 212), (188, 55, 255), (126, 126, 126),
                  (255, 129, 236), (7, 168, 225), (158, 172, 255), (93, 248, 133), (255, 255, 255)))


class IIGSPalette(BasePalette):
    """BMP2DHGR's KEGS32 palette (palette.py:57-78)."""
    ID = Palette.IIGS
    RGB = _table(((0, 0, 0), (221, 0, 51), (136, 85, 34), (255, 102, 0), (0, 119, 0), (85, 85, 85),
                  (0, 221, 0), (255, 255, 0), (0, 0, 153), (221, 0, 221), (170, 170, 170),
                  (255, 153, 136), (34, 34, 255), (102, 170, 255), (0, 255, 153), (255, 255, 255)))


PALETTES = {
    Palette.IIGS: IIGSPalette,
    Palette.NTSC: NTSCPalette
}  # type: Dict[Palette, Type[BasePalette]]
