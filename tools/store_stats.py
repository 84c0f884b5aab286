"""Value range of the store tables (decides whether the 10-bit repack applies)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ii-vision_amd", "transcoder"))
import torch
import _iiv_native as native
import palette
for pal_name, pal in (("NTSC", palette.NTSCPalette), ("IIGS", palette.IIGSPalette)):
    _, dm = native.cie2000_matrix(pal.rgb_array())
    for mode, name in ((native.DHGR, "DHGR"), (native.HGR, "HGR")):
        st = native.build_store_table(mode, dm)
        t = st if isinstance(st, torch.Tensor) else st.tensor
        t = t.view(torch.int16).to(torch.int32) & 0xffff
        print(pal_name, name, "entries", t.numel(), "max", int(t.max()), "mean %.1f" % float(t.float().mean()),
              "frac<256 %.3f" % float((t < 256).float().mean()), "frac<1024 %.4f" % float((t < 1024).float().mean()))
