"""MI355X-native ][-Vision transcode hot path.

The directory name mirrors the project name and is not a Python identifier; load it
with importlib.import_module("ii-vision_amd").  Importing it puts
ii-vision_amd/transcoder on sys.path, which -- exactly like the reference's
transcoder/ directory -- is a flat namespace of top-level modules:

    video, screen, make_data_tables, colours, palette, video_mode   (reference names)
    stream_batch                                                   (batched driver)
    _iiv_native                                                    (ctypes binding of libiivision.so)
"""

import os
import sys

TRANSCODER_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "transcoder")
if TRANSCODER_DIR not in sys.path:
    sys.path.insert(0, TRANSCODER_DIR)

import _iiv_native as native  # noqa: E402

__all__ = ["native", "TRANSCODER_DIR"]
