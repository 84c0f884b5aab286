"""Diagnostic: per-phase s_memtime stamps of prologue_kernel (needs the -DIIV_STAMPS build:
    make -C ii-vision_amd/csrc ../libiivision_stamps.so
    IIV_LIB=$PWD/ii-vision_amd/libiivision_stamps.so python tools/prologue_stamps.py)."""
import os, sys
sys.path.insert(0, os.path.join(os.getcwd(), 'ii-vision_amd', 'transcoder'))
import numpy as np, torch
import _iiv_native as native, stream_batch, palette
_, dm = native.cie2000_matrix(palette.NTSCPalette.rgb_array())
mode = native.HGR if "HGR" in sys.argv[1:] else native.DHGR
IMG = "img" in sys.argv[1:]      # picture-like input (S-img) instead of S-iid
RGB = "rgb" in sys.argv[1:]      # ordered-dither frames of the synthetic RGB clips, a scene cut in front of the last frame (bench.py's e2e legs)
table = native.build_table(mode, dm, True); store = native.build_store_table(mode, dm)
for S in (1, 512, 4096):
    if RGB:
        D = min(S, 256)
        rgb = stream_batch.synth_rgb_torch(D + 1, 4, seed=10)
        rgb[:D, 3] = rgb[1:, 3].clone()       # frame 3 of clip c is clip c + 1's: a scene cut
        m = torch.empty((D, 4, 32, 256), dtype=torch.uint8, device="cuda")
        a = torch.empty((D, 4, 32, 256), dtype=torch.uint8, device="cuda") if mode == native.DHGR else None
        native.frames_to_memory_maps(mode, palette.NTSCPalette.rgb_array(), rgb[:D].contiguous().view(D * 4, 192, 280, 3), 32, out=(m, a))
        reps = (S + D - 1) // D
        fm = m.repeat(reps, 1, 1, 1)[:S].contiguous()
        fa = a.repeat(reps, 1, 1, 1)[:S].contiguous() if a is not None else None
    else:
        fm, fa = (stream_batch.synth_frames_img(S, 4, mode == native.DHGR, seed=5) if IMG else
                  stream_batch.synth_frames_torch(S, 4, mode == native.DHGR, seed=5))
    b = stream_batch.StreamBatch(mode, table, store, S, seeds=[(i+1,i+1) for i in range(S)], dm=dm)
    b.encode_frames(fm, fa, 3)
    ops, segs = b.encode_frames(fm, fa, 1)
    b.enc.check()
    full = np.stack([b.enc.get_state(100, i) for i in range(0, S, max(1, S//16))]).astype(np.int64)
    b.enc.encode(fm, fa, [(3, 0, 1, 292), (3, 1 if mode == native.DHGR else 0, 1, 0)])   # closed generator -> prefix sort
    b.enc.check()
    rows = np.stack([b.enc.get_state(100, i) for i in range(0, S, max(1, S//16))]).astype(np.int64)
    sub = rows[:, [5, 8, 9, 6]]
    print("   ordering sub-phases (init, scatter, rank+store):", np.median(np.diff(sub, axis=1), axis=0).astype(int).tolist())
    st = rows[:, [0, 10, 11, 1, 12]]
    print("   stage sub-phases (cur|tgt loads, mt+cost lut loads, string lut+barrier, MT wave):", np.median(np.diff(st, axis=1), axis=0).astype(int).tolist())
    d = np.diff(rows[:, :8], axis=1)
    print("S=%d last prologue phases (median cycles of s_memtime @100MHz?):" % S)
    print("  prefix: stage,phase1(dw),scan+select,MT,keys+compact,sort,write :", np.median(d, axis=0).astype(int).tolist(), "total", int(np.median(rows[:,7]-rows[:,0])))
    if RGB and S <= 512:      # the slowest streams of the prefix-mode call: a launch ends when its slowest workgroup does
        allr = np.stack([b.enc.get_state(100, i) for i in range(S)]).astype(np.int64)
        tot = allr[:, 7] - allr[:, 0]
        worst = np.argsort(tot)[::-1][:4]
        for w in worst:
            print("   slowest: stream %d total %d clocks; phases %s; ordering sub-phases %s" % (w, tot[w], np.diff(allr[w, :8]).tolist(), np.diff(allr[w, [5, 8, 9, 6]]).tolist()))
    d = np.diff(full[:, :8], axis=1)
    print("  full  : stage,phase1(dw),scan+select,MT,keys+compact,sort,write :", np.median(d, axis=0).astype(int).tolist(), "total", int(np.median(full[:,7]-full[:,0])))
    b.close()
