import os, sys, time
sys.path.insert(0, "/root/repo/ii-vision_amd/transcoder")
import torch, _iiv_native as native, stream_batch, palette
_, dm = native.cie2000_matrix(palette.NTSCPalette.rgb_array())
for mode in (native.DHGR, native.HGR):
    table = native.build_table(mode, dm, True); store = native.build_store_table(mode, dm)
    for kind in ("iid", "img"):
        fm, fa = (stream_batch.synth_frames_img(1, 160, mode == native.DHGR, seed=99) if kind == "img" else
                  stream_batch.synth_frames_torch(1, 160, mode == native.DHGR, seed=99))
        for kern in ("team", True):
            b = stream_batch.StreamBatch(mode, table, store, 1, seeds=[(1, 1)], dm=dm, fourth_offset=True)
            b.enc.set_greedy_kernel(kern)
            b.encode_frames(fm, fa, 10); torch.cuda.synchronize()
            dts = []
            for _ in range(3):
                t0 = time.perf_counter(); b.encode_frames(fm, fa, 50); torch.cuda.synchronize(); dts.append(time.perf_counter() - t0)
            b.enc.check(); b.close()
            print("%s S-%s fourth offset, kernel %s: %.0f frames/s" % ("DHGR" if mode == native.DHGR else "HGR", kind, kern, 50 / sorted(dts)[1]))
