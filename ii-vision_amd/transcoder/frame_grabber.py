"""Frames of a video as Apple II memory maps -- converted on the GPU.

Mirrors the interface of the reference's transcoder/frame_grabber.py: FrameGrabber carries
`video_mode` and `input_frame_rate` (video.Video reads the latter, video.py:31-33) and
`frames()` yields (main, aux) screen.MemoryMap pairs exactly as FileFrameGrabber.frames() does
(frame_grabber.py:56-147), aux being None for HGR.

What differs, and why: the reference decodes with ffmpeg, resizes each frame to 280x192 with
PIL (frame_grabber.py:75,100) and shells out to the external tool /usr/local/bin/bmp2dhr for
the image -> memory-map conversion (frame_grabber.py:78-82,103-108).  Decoding and resizing
stay out of scope (they are not on the transcode hot path); the conversion itself is row f3
of SURVEY 8f and runs here as a HIP kernel (csrc/iiv_ingest.hip).  bmp2dhr is not part of the
reference's source, so its output cannot be matched: the conversion is specified in
include/iivision.h (iiv_frames_to_memory_maps) and the tests hold the kernel to it.
"""

from typing import Iterator, Tuple

import numpy as np

import _iiv_native as native
import palette as palette_mod
import screen
from palette import Palette
from video_mode import VideoMode


class FrameGrabber:
    """frame_grabber.py:18-24."""

    def __init__(self, mode: VideoMode):
        self.video_mode = mode
        self.input_frame_rate = 30

    def frames(self) -> Iterator[Tuple[screen.MemoryMap, screen.MemoryMap]]:
        raise NotImplementedError


class ArrayFrameGrabber(FrameGrabber):
    """Frames given as an array (n, 192, 280, 3) uint8 -- what FileFrameGrabber holds after its
    resize -- converted `batch` frames at a time by iiv_frames_to_memory_maps."""

    def __init__(self, frames_rgb, mode: VideoMode, palette: Palette = Palette.NTSC, dither: int = 32,
                 input_frame_rate: float = 30, batch: int = 256):
        super().__init__(mode)
        rgb = np.asarray(frames_rgb)
        if rgb.dtype != np.uint8 or rgb.ndim != 4 or rgb.shape[1:] != (192, 280, 3):
            raise ValueError("frames must be uint8 (n, 192, 280, 3) (frame_grabber.py:75: 280x192 RGB)")
        self._rgb = rgb
        self.palette = palette
        # 0..255: amplitude of the 4x4 ordered dither; "diffusion" (= native.DITHER_DIFFUSION): Floyd-Steinberg error
        # diffusion, the kind of dither the reference asks bmp2dhr for (D9, frame_grabber.py:80-82,106-108)
        self.dither = native.DITHER_DIFFUSION if dither == "diffusion" else int(dither)
        self.input_frame_rate = input_frame_rate
        self.batch = int(batch)

    def memory_maps(self, first=0, count=None):
        """(main, aux) CUDA uint8 tensors (count, 32, 256) of frames first .. first + count - 1:
        the form stream_batch.StreamBatch consumes, without a trip through host memory maps."""
        import torch
        count = len(self._rgb) - first if count is None else count
        rgb = torch.from_numpy(np.ascontiguousarray(self._rgb[first:first + count])).cuda()
        mode = native.DHGR if self.video_mode == VideoMode.DHGR else native.HGR
        pal = palette_mod.PALETTES[self.palette].rgb_array()
        return native.frames_to_memory_maps(mode, pal, rgb, self.dither)

    def frames(self):
        for first in range(0, len(self._rgb), self.batch):
            main, aux = self.memory_maps(first, min(self.batch, len(self._rgb) - first))
            main = main.cpu().numpy()
            aux = aux.cpu().numpy() if aux is not None else None
            for i in range(main.shape[0]):
                yield (screen.MemoryMap(screen_page=1, page_offset=main[i].copy()),
                       screen.MemoryMap(screen_page=1, page_offset=aux[i].copy()) if aux is not None else None)
