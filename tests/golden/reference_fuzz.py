#!/usr/bin/env python3
"""Randomised differential run of the ORACLE against the imported REFERENCE (build container only: it needs
/root/reference; nothing here runs at test time, nothing of the reference is stored).  The golden vectors of
make_golden.py pin the oracle on a dozen fixed cases; this script throws random cases at the pair -- input kinds (S-iid,
S-coh, converging, picture-like), modes, palettes, seeds, ragged schedules with abandoned and one-opcode generators, runs
to exhaustion, and (--fourth) the reference with its exit test `len(offsets) == 3` reading 4 against the oracle's
fourth-offset flag -- and compares every opcode, the final memory maps, priorities, packed screen, out_of_work and both
RNG positions.   python tests/golden/reference_fuzz.py [rounds] [--fourth]     (log: profiles/r03_reference_fuzz.txt)"""
import contextlib
import ctypes as C
import io
import os
import random
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import make_golden as MG  # noqa: E402  (setup_reference: the Appendix-B import shim)

rounds = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 40
fourth = "--fourth" in sys.argv
import oracle as O  # noqa: E402
O.build()
MG.setup_reference("/tmp/iiv_ref")
MG.write_reference_tables("/tmp/iiv_ref", O)
import frame_grabber  # noqa: E402
import palette  # noqa: E402
import screen  # noqa: E402
import video  # noqa: E402
import video_mode  # noqa: E402

if fourth:
    import inspect
    import textwrap
    src = textwrap.dedent(inspect.getsource(video.Video._index_changes))
    assert src.count("if len(offsets) == 3:") == 1
    ns = dict(vars(video))
    exec(compile(src.replace("if len(offsets) == 3:", "if len(offsets) == 4:"), "<_index_changes, exit test at 4>", "exec"), ns)
    video.Video._index_changes = ns["_index_changes"]

rng = np.random.default_rng(int(os.environ.get("IIV_FUZZ_SEED", "31")))
tables = {}
t_start, total = time.time(), 0
for rnd in range(rounds):
    mode = int(rng.integers(0, 2))
    mode_name = "DHGR" if mode else "HGR"
    pal = 5 if rng.random() < 0.7 else 0
    kind = ("iid", "coh", "static", "img")[int(rng.integers(0, 4))]
    nf = 3
    hi = 128 if mode else 256
    if kind == "img":   # dithered moving bars, packed 7 dots per byte (as stream_batch.synth_frames_img)
        frames = np.zeros((nf, 2, 32, 256), np.uint8)
        W = 560 if mode else 280
        period, speed, slope, phase = int(rng.integers(24, 120)), int(rng.integers(1, 6)), int(rng.integers(-2, 3)), int(rng.integers(0, 120))
        bayer = np.array([[0, 8, 2, 10], [12, 4, 14, 6], [3, 11, 1, 9], [15, 7, 13, 5]])
        y, x = np.mgrid[0:192, 0:W]
        for f in range(nf):
            dots = (((x + slope * y + speed * f + phase) % period) * 17) // period > bayer[y % 4, x % 4]
            by = (dots.reshape(192, W // 7, 7) * (1 << np.arange(7))).sum(-1).astype(np.uint8)
            for yy in range(192):
                for xx in range(40):
                    p, o = int(screen.X_Y_TO_PAGE[yy, xx]), int(screen.X_Y_TO_OFFSET[yy, xx])
                    if mode:
                        frames[f, 1, p, o], frames[f, 0, p, o] = by[yy, 2 * xx], by[yy, 2 * xx + 1]
                    else:
                        frames[f, 0, p, o] = by[yy, xx]
    else:
        frames = np.zeros((nf, 2, 32, 256), np.uint8)
        keep_p = {"iid": 0.0, "coh": 0.9, "static": 0.98}[kind]
        for b in range(2 if mode else 1):
            for f in range(nf):
                new = rng.integers(0, hi, (32, 256), dtype=np.uint8)
                if f and keep_p:
                    new = np.where(rng.random((32, 256)) < keep_p, frames[f - 1, b], new)
                new[screen.SCREEN_HOLES] = 0
                frames[f, b] = new
    sched = []
    exhaust = rng.random() < 0.25
    for _ in range(int(rng.integers(2, 7))):
        k = int(rng.choice([1, 1, 2, 3, 40, 183, 292, 490, 900])) if not exhaust else int(rng.integers(2500, 7000))
        sched.append((int(rng.integers(0, nf)), int(rng.integers(0, 2)) if mode else 0, k))
    sp, sn = int(rng.integers(1 << 20)), int(rng.integers(1 << 20))
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        ref = MG.run_reference(mode_name, pal, frames[:, :2 if mode else 1], sched, sp, sn)
    key = (mode, pal)
    if key not in tables:
        tables[key] = O.build_table(mode, O.cie2000_matrix(O.PALETTE_RGB[pal])[1], symmetric=True)
    for structured in (False, True):
        v = O.Video(mode, tables[key], seed_py=sp, seed_np=sn)
        v.set_fourth_offset(fourth)
        out = []
        for fi, ia, n in sched:
            v.encode_frame(frames[fi, 0], frames[fi, 1] if mode else None, ia)
            out.append(v.next(int(n), structured=structured))
        out = np.concatenate(out)
        tag = (rnd, mode_name, pal, kind, structured, sched)
        assert (out == ref["ops"]).all(), ("ops",) + tag
        assert (v.memory(0) == ref["mem_main"]).all() and (v.update_priority(0) == ref["up_main"]).all(), ("state",) + tag
        assert (v.packed == ref["packed"]).all(), ("packed",) + tag
        if mode:
            assert (v.memory(1) == ref["mem_aux"]).all() and (v.update_priority(1) == ref["up_aux"]).all(), ("aux",) + tag
        assert [int(v.out_of_work(0)), int(v.out_of_work(1))] == ref["out_of_work"].tolist(), ("out_of_work",) + tag
        rp, rn = v.rng_py(), v.rng_np()
        L = O.lib()
        assert [L.orc_py_getrandbits8(C.byref(rp)) for _ in range(4)] == ref["py_next"].tolist(), ("py rng",) + tag
        assert [L.orc_np_randint256(C.byref(rn)) for _ in range(4)] == ref["np_next"].tolist(), ("np rng",) + tag
    total += len(ref["ops"])
    print("round %2d ok: %s pal=%d S-%s%s ops=%s" % (rnd, mode_name, pal, kind, " (to exhaustion)" if exhaust else "", [s[2] for s in sched]), flush=True)
print("reference fuzz%s: %d rounds, %d opcodes, oracle (heap form and restructured form) == reference in every one (%.0f s)" % (
    " with the exit test at 4" if fourth else "", rounds, total, time.time() - t_start))
