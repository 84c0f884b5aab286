"""How many 3-value words of the HGR store table fit 10 bits + a 2-bit per-word base?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ii-vision_amd", "transcoder"))
import torch
import _iiv_native as native
import palette
for pal_name, pal in (("NTSC", palette.NTSCPalette), ("IIGS", palette.IIGSPalette)):
    _, dm = native.cie2000_matrix(pal.rgb_array())
    for mode, name, bits in ((native.HGR, "HGR", 14), (native.DHGR, "DHGR", 13)):
        st = native.build_store_table(mode, dm)
        t = st if isinstance(st, torch.Tensor) else st.tensor
        t = (t.view(torch.int16).to(torch.int32) & 0xffff).view(-1, 1 << bits)
        n = ((1 << bits) + 2) // 3
        pad = torch.zeros((t.shape[0], 3 * n - (1 << bits)), dtype=t.dtype, device=t.device)
        w = torch.cat([t, pad], 1).view(t.shape[0], n, 3)
        w = w[:, :-1]  # (ignore the padded last word)
        mn, mx = w.min(-1).values, w.max(-1).values
        for step in (256, 128):
            base = torch.clamp(mn // step, max=3) * step
            ok = (mx - base) <= 1022
            print(pal_name, name, "base step", step, "words not representable: %.4f %%" % (100 * (1 - ok.float().mean().item())),
                  "values escaping (plain 10 bit): %.2f %%" % (100 * (w >= 1023).float().mean().item()))
