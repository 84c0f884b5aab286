"""Diagnostic: where a wave of greedy_wave_kernel spends its shader clocks (-DIIV_STAMPS build):
    make -C ii-vision_amd/csrc ../libiivision_stamps.so
    IIV_LIB=$PWD/ii-vision_amd/libiivision_stamps.so python tools/greedy_phases.py [streams]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ii-vision_amd", "transcoder"))
import numpy as np, torch
import _iiv_native as native, stream_batch, palette
S = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
KIND = sys.argv[2] if len(sys.argv) > 2 else "iid"      # iid | coh | img
_, dm = native.cie2000_matrix(palette.NTSCPalette.rgb_array())
mode = native.DHGR
table = native.build_table(mode, dm, True); store = native.build_store_table(mode, dm)
if KIND == "img":
    fm, fa = stream_batch.synth_frames_img(S, 6, True, seed=5)
else:
    fm, fa = stream_batch.synth_frames_torch(S, 6, True, seed=5, coherent=KIND == "coh")
b = stream_batch.StreamBatch(mode, table, store, S, seeds=[(i + 1, i + 1) for i in range(S)], dm=dm)
b.enc.set_greedy_kernel(True)
b.encode_frames(fm, fa, 4)
b.enc.encode(fm, fa, [(4, 0, 1, 292)])     # the measured launch: 292 opcodes per stream
b.enc.check()
allst = np.stack([b.enc.get_state(100, i) for i in range(0, S, max(1, S // 512))]).astype(np.int64)
rows, t0, t1, ops = allst[:, 16:20], allst[:, 24], allst[:, 25], allst[:, 26]
print("S-%s: %.1f list entries through the pipeline per opcode emitted (the rest were dead when their turn came)" % (
    KIND, allst[:, 27].sum() / max(ops.sum(), 1)))
span = t1.max() - t0.min()
life = t1 - t0
print("S=%d: wave lifetimes mean %.0f  min %.0f  max %.0f clocks; launch span %.0f -> mean lifetime / span %.2f" % (
    S, life.mean(), life.min(), life.max(), span, life.mean() / span))
print("        starts: first %.0f, median %.0f, last %.0f clocks after the first wave" % (
    0, np.median(t0 - t0.min()), (t0 - t0.min()).max()))
names = ["8 loads + take + row request", "MT19937 generation", "table-word wait + score + apply", "waiting for the row (requested two entries earlier)"]
tot = rows.sum(axis=1).mean()
print("        clocks per opcode %.0f:" % (tot / ops.mean()))
for n, v in zip(names, rows.mean(axis=0)):
    print("          %-44s %7.0f  (%4.1f %%)" % (n, v / ops.mean(), 100 * v / tot))
