"""Diagnostic: frames/s by diff-weight mode (split table / recurrence / full-table gather) at several
batch sizes.  python tools/dw_mode_probe.py [HGR]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.getcwd(), 'ii-vision_amd', 'transcoder'))
import torch
import _iiv_native as native, stream_batch, palette
mode = native.HGR if len(sys.argv) > 1 and sys.argv[1] == "HGR" else native.DHGR
_, dm = native.cie2000_matrix(palette.NTSCPalette.rgb_array())
table = native.build_table(mode, dm, True); store = native.build_store_table(mode, dm)
for S in (1, 64, 512, 4096):
    fm, fa = stream_batch.synth_frames_torch(S, 40, mode == native.DHGR, seed=5)
    for dw in ("split", "recurrence", "table"):
        b = stream_batch.StreamBatch(mode, table, store, S, seeds=[(i + 1, i + 1) for i in range(S)], dm=dm)
        b.enc.set_diff_weights_mode(dw)
        b.encode_frames(fm, fa, 8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        b.encode_frames(fm, fa, 30)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        b.enc.check()
        b.close()
        print("S=%5d  %-10s  %10.0f frames/s" % (S, dw, S * 30 / dt), flush=True)
