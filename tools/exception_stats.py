"""Diagnostic (DHGR): how many (content, window) pairs the narrow table form sends to the dense table
-- an exception-mask bit covers 512 pairs -- against how many really differ from l1 + r1."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ii-vision_amd", "transcoder"))
import torch
import _iiv_native as native, palette
for pal in (palette.NTSCPalette, palette.IIGSPalette):
    _, dm = native.cie2000_matrix(pal.rgb_array())
    mode = native.DHGR
    dense = native.build_store_table(mode, dm).view(-1).to(torch.int64)
    left, right, _ = native.build_split_store_table(mode, dm, expanded=False)
    l1 = (left.to(torch.int64) & 0xffffffff) >> 16
    r1 = (right.to(torch.int64) & 0xffffffff) >> 16
    idx = torch.arange(dense.numel(), device="cuda")
    win, c, o = idx & 8191, (idx >> 13) & 127, idx >> 20
    v = l1[(((o << 5) | (c & 31)) << 8) | (win & 255)] + r1[(((o << 6) | ((c >> 1) & 63)) << 9) | (win >> 4)]
    differ = int((v != dense).sum())
    _, n_exc = native.build_narrow_store_table(mode, dm, native.build_store_table(mode, dm))
    print("%s: %d pairs; l1 + r1 differs from S for %d (%.3f %%); the masks send %d (%.3f %%) to the dense table" % (
        pal.__name__, dense.numel(), differ, 100.0 * differ / dense.numel(), n_exc, 100.0 * n_exc / dense.numel()))
