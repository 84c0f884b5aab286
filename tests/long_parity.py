"""Long-horizon differential run: a few whole 1000-frame clips (BASELINE configs 3 / 4: DHGR and HGR,
Movie pacing, bank flips), GPU through the C ABI against the oracle -- every opcode, the final
screen, priorities and both RNG positions.  The oracle here is the checker.  On an MI355X:
    python tests/long_parity.py [frames] [clips] [auto|wave] [palette id] [iid|coh|img]   (about a minute at 1000 x 8;
few clips run in the eight-waves-per-clip team kernel unless "wave" asks for the one-wave kernel)"""
import concurrent.futures
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "ii-vision_amd", "transcoder")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import oracle as O  # noqa: E402
import _iiv_native as native  # noqa: E402
import stream_batch  # noqa: E402

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
kernel = sys.argv[3] if len(sys.argv) > 3 else "auto"
pal = int(sys.argv[4]) if len(sys.argv) > 4 else 5        # 5 = NTSC, 0 = //gs (palette.py:18-23)
kind = sys.argv[5] if len(sys.argv) > 5 else "iid"        # iid | coh | img
O.build()
dm = O.cie2000_matrix(O.PALETTE_RGB[pal])[1]
for mode in (native.DHGR, native.HGR):
    t0 = time.time()
    otab = O.build_table(mode, dm, symmetric=True)
    table, store = native.build_table(mode, dm, True), native.build_store_table(mode, dm)
    if kind == "img":
        fm, fa = stream_batch.synth_frames_img(n, n_frames, mode == native.DHGR, seed=77)
    else:
        fm, fa = stream_batch.synth_frames_torch(n, n_frames, mode == native.DHGR, seed=77, coherent=kind == "coh")
    seeds = [(i + 1, 100 + i) for i in range(n)]
    b = stream_batch.StreamBatch(mode, table, store, n, seeds=seeds, dm=dm)
    b.enc.set_greedy_kernel(True if kernel == "wave" else None)
    got, segs = [], []
    for start in range(0, n_frames, 50):     # the driver's 50-frame steps, generators continued across calls
        ops, s = b.encode_frames(fm, fa, min(50, n_frames - start))
        got.append(ops.cpu().numpy())
        segs += s
    b.enc.check()
    got = np.concatenate(got, axis=1)
    fmh, fah = fm.cpu().numpy(), (fa.cpu().numpy() if fa is not None else None)

    def run(i):
        v = O.Video(mode, otab, seed_py=seeds[i][0], seed_np=seeds[i][1])
        out = []
        for (fr, ia, restart, k) in segs:
            if restart:
                v.encode_frame(fmh[i, fr], fah[i, fr] if fah is not None else None, ia)
            if k:
                out.append(v.next(k))
        return v, np.concatenate(out)

    with concurrent.futures.ThreadPoolExecutor(n) as ex:
        res = list(ex.map(run, range(n)))
    for i, (v, exp) in enumerate(res):
        bad = np.nonzero((got[i] != exp).any(axis=1))[0]
        assert len(bad) == 0, ("opcodes", mode, i, int(bad[0]))
        assert (b.enc.get_state(native.STATE_MEM_MAIN, i) == v.memory(0)).all()
        assert (b.enc.get_state(native.STATE_UP_MAIN, i) == v.update_priority(0)).all()
        if mode == native.DHGR:
            assert (b.enc.get_state(native.STATE_MEM_AUX, i) == v.memory(1)).all()
            assert (b.enc.get_state(native.STATE_UP_AUX, i) == v.update_priority(1)).all()
        cnt = b.enc.get_state(native.STATE_COUNTERS, i)
        assert (int(cnt[0]), int(cnt[1])) == v.draws()
    print("%s, %s kernel, palette %d, S-%s: %d clips x %d frames = %d opcodes each, all equal (%.0f s)" % (
        "DHGR" if mode == native.DHGR else "HGR", kernel, pal, kind, n, n_frames, got.shape[1], time.time() - t0), flush=True)
    b.close()
