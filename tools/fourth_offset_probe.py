"""What IIV_OPT_FOURTH_OFFSET buys and costs (f4; not the reference's stream): for the same clips and the same opcode
budget, the perceptual error left between screen and target after every frame (sum of Bitmap.diff_weights over both
banks), with the reference's two extra offsets and with three -- and how many of an opcode's four stores are distinct.
    python tools/fourth_offset_probe.py [DHGR|HGR] [iid|coh|img] [clips] [frames]
(the many-stream rates: bench.py --fourth)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ii-vision_amd", "transcoder"))
import numpy as np, torch
import _iiv_native as native, stream_batch, palette

MODE = sys.argv[1] if len(sys.argv) > 1 else "DHGR"
KIND = sys.argv[2] if len(sys.argv) > 2 else "coh"
S = int(sys.argv[3]) if len(sys.argv) > 3 else 32
F = int(sys.argv[4]) if len(sys.argv) > 4 else 30
mode = native.DHGR if MODE == "DHGR" else native.HGR
dhgr = mode == native.DHGR
_, dm = native.cie2000_matrix(palette.NTSCPalette.rgb_array())
table = native.build_table(mode, dm, True)
store = native.build_store_table(mode, dm)
if KIND == "img":
    fm, fa = stream_batch.synth_frames_img(S, F, dhgr, seed=11)
else:
    fm, fa = stream_batch.synth_frames_torch(S, F, dhgr, seed=11, coherent=KIND == "coh")


def screen_error(enc, frame):
    mem_m = np.stack([enc.get_state(native.STATE_MEM_MAIN, i) for i in range(S)])
    mem_a = np.stack([enc.get_state(native.STATE_MEM_AUX, i) for i in range(S)]) if dhgr else None
    src = native.pack(mode, mem_m, mem_a)
    tgt = native.pack(mode, fm[:, frame].cpu().numpy(), fa[:, frame].cpu().numpy() if dhgr else None)
    return sum(int(native.diff_weights(mode, table, src[i], tgt[i], ia).sum()) for i in range(S) for ia in ((0, 1) if dhgr else (0,)))


res = {}
for fourth in (False, True):
    b = stream_batch.StreamBatch(mode, table, store, S, seeds=[(i + 1, i + 1) for i in range(S)], dm=dm, fourth_offset=fourth)
    errs, distinct = [], []
    for f in range(F):
        ops, _ = b.encode_frames(fm, fa, 1)
        b.enc.check()
        o = ops.cpu().numpy().reshape(-1, 6)[:, 2:6]
        distinct.append(np.mean([(np.sort(o, axis=1)[:, 1:] != np.sort(o, axis=1)[:, :-1]).sum(axis=1) + 1]))
        errs.append(screen_error(b.enc, f))
    res[fourth] = (np.array(errs, np.float64), float(np.mean(distinct)))
    b.close()
e0, e1 = res[False][0], res[True][0]
print("%s S-%s, %d clips x %d frames, 490 opcodes per frame:" % (MODE, KIND, S, F))
print("  distinct offsets per opcode: %.2f (reference: two extra offsets + a copy of the first) -> %.2f (fourth offset)" % (
    res[False][1], res[True][1]))
for f in sorted(set([0, 1, 2, 4, 9, F // 2, F - 1])):
    print("  error left after frame %3d: %12.0f -> %12.0f  (%.1f %% less)" % (f, e0[f], e1[f], 100.0 * (1 - e1[f] / e0[f])))
print("  mean over the last %d frames: %.1f %% less error left on the screen per frame" % (
    F - F // 3, 100.0 * (1 - e1[F // 3:].mean() / e0[F // 3:].mean())))
