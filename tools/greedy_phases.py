"""Diagnostic: where a wave of greedy_wave_kernel spends its cycles (-DIIV_STAMPS build):
    make -C ii-vision_amd/csrc ../libiivision_stamps.so
    IIV_LIB=$PWD/ii-vision_amd/libiivision_stamps.so python tools/greedy_phases.py [streams]"""
import os, sys
sys.path.insert(0, os.path.join(os.getcwd(), 'ii-vision_amd', 'transcoder'))
import numpy as np, torch
import _iiv_native as native, stream_batch, palette
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
_, dm = native.cie2000_matrix(palette.NTSCPalette.rgb_array())
mode = native.DHGR
table = native.build_table(mode, dm, True); store = native.build_store_table(mode, dm)
fm, fa = stream_batch.synth_frames_torch(S, 6, True, seed=5)
b = stream_batch.StreamBatch(mode, table, store, S, seeds=[(i+1,i+1) for i in range(S)], dm=dm)
b.encode_frames(fm, fa, 4)
b.enc.encode(fm, fa, [(4, 0, 1, 292)])     # the measured launch: 292 opcodes per stream
b.enc.check()
allst = np.stack([b.enc.get_state(100, i) for i in range(0, S, max(1, S // 256))]).astype(np.int64)
rows = allst[:, 16:24]
t0, t1 = allst[:, 24], allst[:, 25]
span = t1.max() - t0.min()
print("wave lifetimes (loop only): mean %.0f  min %.0f  max %.0f cycles; launch span %.0f  -> mean/span %.2f; starts spread %.0f" % (
    (t1 - t0).mean(), (t1 - t0).min(), (t1 - t0).max(), span, (t1 - t0).mean() / span, t0.max() - t0.min()))
names = ["loop/pushed", "form chunk+rows", "rows wait+issue loads", "next chunk", "twist", "values wait", "score+apply", "-"]
tot = rows.sum(axis=1).mean()
print("S=%d: cycles per wave for 292 opcodes: %.0f (%.0f per opcode)" % (S, tot, tot / 292))
for n, v in zip(names, rows.mean(axis=0)):
    print("  %-24s %9.0f  %5.1f %%  (%.0f / opcode)" % (n, v, 100 * v / tot, v / 292))
