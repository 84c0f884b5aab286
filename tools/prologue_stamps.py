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
table = native.build_table(mode, dm, True); store = native.build_store_table(mode, dm)
for S in (1, 512, 4096):
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
    d = np.diff(full[:, :8], axis=1)
    print("  full  : stage,phase1(dw),scan+select,MT,keys+compact,sort,write :", np.median(d, axis=0).astype(int).tolist(), "total", int(np.median(full[:,7]-full[:,0])))
    b.close()
