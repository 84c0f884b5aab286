"""Diagnostic: the joint content choice (f4) run far past the sorted list -- into the re-queued bag and out of work --
against the oracle's definition.   python tools/joint_exhaust_check.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "ii-vision_amd", "transcoder")):
    sys.path.insert(0, p)
import numpy as np, torch
import oracle as O, _iiv_native as native, stream_batch
O.build()
dm = O.cie2000_matrix(O.PALETTE_RGB[5])[1]
bad = 0
for mode in (1, 0):
    otab = O.build_table(mode, dm, symmetric=True)
    t, s = native.build_table(mode, dm, True), native.build_store_table(mode, dm)
    n, nf = 2, 2
    fm, fa = stream_batch.synth_frames_torch(n, nf, mode == 1, seed=5, coherent=True, device="cpu", keep=0.97)
    sched = [(0, 0, 1, 9000), (1, 0, 1, 2500)]
    enc = native.Encoder(mode, t, s, n, dm=dm)
    enc.set_content_choice(True)
    seeds = [(i + 3, i + 9) for i in range(n)]
    for i, (sp, sn) in enumerate(seeds):
        enc.set_state(native.STATE_RNG_PY, O.mt_seed_py(sp).state_words(), i)
        enc.set_state(native.STATE_RNG_NP, O.mt_seed_np(sn).state_words(), i)
    got = enc.encode(fm.cuda(), fa.cuda() if fa is not None else None, sched).cpu().numpy()
    enc.check()
    for i in range(n):
        t0 = time.time()
        v = O.Video(mode, otab, seed_py=seeds[i][0], seed_np=seeds[i][1])
        v.set_joint(True)
        exp = []
        for (fr, a, restart, k) in sched:
            if restart:
                v.encode_frame(fm[i, fr].numpy(), fa[i, fr].numpy() if fa is not None else None, a)
            exp.append(v.next(k))
        exp = np.concatenate(exp)
        mism = np.nonzero((got[i] != exp).any(axis=1))[0]
        print("mode %d stream %d: %d ops, %d pads, first mismatch %s (%.0f s)" % (
            mode, i, len(exp), int((exp[:, 0] == 32).sum() if False else 0), mism[0] if len(mism) else None, time.time() - t0), flush=True)
        bad += len(mism) > 0
    enc.close()
print("BAD" if bad else "ALL EQUAL")
