import os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ii-vision_amd", "transcoder"))
import torch
import _iiv_native as native, stream_batch, palette
for mode in (native.DHGR, native.HGR):
    _, dm = native.cie2000_matrix(palette.NTSCPalette.rgb_array())
    table = native.build_table(mode, dm, True); store = native.build_store_table(mode, dm)
    for kind in ("iid", "img"):
        if kind == "img":
            fm, fa = stream_batch.synth_frames_img(1, 120, mode == native.DHGR, seed=99)
        else:
            fm, fa = stream_batch.synth_frames_torch(1, 120, mode == native.DHGR, seed=99)
        for kern in ("team", True):
            b = stream_batch.StreamBatch(mode, table, store, 1, seeds=[(1, 1)], dm=dm)
            b.enc.set_greedy_kernel(kern)
            b.encode_frames(fm, fa, 10)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            b.encode_frames(fm, fa, 50)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print("%s S-%s kernel %s: %.0f frames/s" % ("DHGR" if mode == native.DHGR else "HGR", kind, kern, 50 / dt), flush=True)
            b.close()
