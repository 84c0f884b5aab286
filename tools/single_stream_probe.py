import sys, time, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/ii-vision_amd/transcoder')
import torch, numpy as np
import _iiv_native as native, stream_batch, palette
for mode in (native.DHGR, native.HGR):
    _, dm = native.cie2000_matrix(palette.NTSCPalette.rgb_array())
    table = native.build_table(mode, dm, True); store = native.build_store_table(mode, dm)
    dhgr = mode == native.DHGR
    for n in (1, 8):
        fm, fa = stream_batch.synth_frames_torch(n, 60, dhgr, seed=99)
        for wave in (False, True, "team"):
            b = stream_batch.StreamBatch(mode, table, store, n, seeds=[(i+1,i+1) for i in range(n)], dm=dm)
            b.enc.set_greedy_kernel(wave)
            b.enc.profile(True)
            b.encode_frames(fm, fa, 10)
            torch.cuda.synchronize()
            t0=time.perf_counter()
            b.encode_frames(fm, fa, 50)
            torch.cuda.synchronize()
            dt=time.perf_counter()-t0
            p=b.enc.profile_read()
            print("mode",mode,"n",n,"wave",wave,"fps/stream %.0f"%(50/dt), "us/op %.3f"%(1e6*dt/(50*490)), "prologue ms %.4f greedy ms %.4f"%(p['prologue_ms']/p['prologue_launches'], p['greedy_ms']/p['greedy_launches']))
            b.close()
