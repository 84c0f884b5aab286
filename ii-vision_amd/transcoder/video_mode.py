"""Video encoding modes (mirrors transcoder/video_mode.py:6-8)."""

import enum


class VideoMode(enum.Enum):
    HGR = 0   # 280x192 hi-res, one 8 KiB bank
    DHGR = 1  # 560x192 double hi-res, MAIN + AUX banks
