"""Diagnostic: distribution of update_priority in steady state (bucket sizing).   python tools/priority_stats.py [HGR] [frames]"""
import os, sys
sys.path.insert(0, os.path.join(os.getcwd(), 'ii-vision_amd', 'transcoder'))
import numpy as np, torch
import _iiv_native as native, stream_batch, palette
_, dm = native.cie2000_matrix(palette.NTSCPalette.rgb_array())
mode = native.HGR if "HGR" in sys.argv[1:] else native.DHGR
DH = mode == native.DHGR
NF = next((int(a) for a in sys.argv[1:] if a.isdigit()), 60)
table = native.build_table(mode, dm, True); store = native.build_store_table(mode, dm)
S = 8
for coh in (False, True, "img"):
    fm, fa = (stream_batch.synth_frames_img(S, NF, DH, seed=5) if coh == "img" else
              stream_batch.synth_frames_torch(S, NF, DH, seed=5, coherent=coh))
    b = stream_batch.StreamBatch(mode, table, store, S, seeds=[(i+1,i+1) for i in range(S)], dm=dm)
    b.encode_frames(fm, fa, NF); b.enc.check()
    for i in (0, 3):
        up = b.enc.get_state(native.STATE_UP_MAIN, i).reshape(-1)
        nz = up[up != 0]
        mx = int(nz.max()); sh = 0 if mx < 1024 else mx.bit_length() - 10
        bins = np.bincount(nz >> sh, minlength=1024)
        srt = np.sort(nz)[::-1]
        need = 876 if DH else 1470     # 3 x the opcodes of a Movie-paced generator
        thr = srt[need] if len(srt) > need else 0
        top = np.sort(bins)[::-1][:6]
        print("   largest buckets:", top.tolist(), " entries in buckets > 96:", int(bins[bins > 96].sum()))
        print("img" if coh == "img" else "coh" if coh else "iid", "stream", i, "n", len(nz), "max", mx, "mean %.0f" % nz.mean(), "p50", int(np.median(nz)),
              "p88", int(np.percentile(nz, 88)), "sh", sh, "max bucket", bins.max(), "bucket at thr", bins[thr >> sh],
              "nonempty", (bins > 0).sum())
    b.close()
