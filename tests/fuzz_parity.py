"""Randomised differential run: GPU (through the C ABI) against the oracle, many seeds,
ragged schedules, all input kinds, both kernels and table forms.  The oracle here is the
checker.  On an MI355X:  python tests/fuzz_parity.py [rounds]   (tests/test_gpu_fuzz.py runs a few
rounds of it inside the GPU suite)"""
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

rounds = int(os.environ.get("IIV_FUZZ_ROUNDS", sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].isdigit() else 12))
rng = np.random.default_rng(int(os.environ.get("IIV_FUZZ_SEED", "2026")))
O.build()
dms = {pal: O.cie2000_matrix(O.PALETTE_RGB[pal])[1] for pal in (5, 0)}
otab, dtab = {}, {}
t_start = time.time()
total_ops = 0
for rnd in range(rounds):
    mode = int(rng.integers(0, 2))
    pal = 5 if rng.random() < 0.7 else 0
    key = (mode, pal)
    if key not in otab:
        otab[key] = O.build_table(mode, dms[pal], symmetric=True)
        dtab[key] = (native.build_table(mode, dms[pal], True), native.build_store_table(mode, dms[pal]))
    n, nf = 8, 4
    kind = ("iid", "coh", "img", "static")[int(rng.integers(0, 4))]   # static: converging content (list -> bag -> out of work)
    if kind == "img":
        fm, fa = stream_batch.synth_frames_img(n, nf, mode == 1, seed=int(rng.integers(1 << 30)), device="cpu")
    else:
        fm, fa = stream_batch.synth_frames_torch(n, nf, mode == 1, seed=int(rng.integers(1 << 30)),
                                                 coherent=kind != "iid", device="cpu", keep=0.98 if kind == "static" else 0.9)
    # schedule: (frame, is_aux, restart, n_ops) with ragged lengths, continued generators, zero-length creations
    def random_schedule(lo=4, hi=10):
        sched, f, ia = [], 0, 0
        for _ in range(int(rng.integers(lo, hi))):
            k = int(rng.choice([0, 1, 2, 63, 64, 65, 127, 200, 292, 490, 700, 1500]))
            restart = 1 if not sched or rng.random() < 0.7 else 0
            if restart:
                f = int(rng.integers(0, nf))
                ia = int(rng.integers(0, 2)) if mode == 1 else 0
            sched.append((f, ia, restart, k))
        return sched
    sched = random_schedule()
    # one round in seven: a marathon -- one generator pulled far past its sorted list, through the re-queued bag and
    # (usually) out of work, then another one on top of what it left
    marathon = bool(rng.random() < 1 / 7)
    if marathon:
        sched = [(int(rng.integers(0, nf)), 0, 1, int(rng.integers(5000, 9500))), (int(rng.integers(0, nf)), int(rng.integers(0, 2)) if mode == 1 else 0, 1, 1200)]
    wave = (True, "shared", "shared", "team", "team", False)[int(rng.integers(0, 6))]
    recurrence = ("split", "split", True, True, False)[int(rng.integers(0, 5))]
    prefix = bool(rng.random() < 0.7)
    # one round in six: the joint content choice (f4) against the oracle's definition of it
    # (128 / 256 times the lookups per step on the CPU: shorter schedules, fewer streams)
    joint = bool(rng.random() < 1 / 6)
    if joint:
        n = 2 if marathon else 3
        fm, fa = fm[:n], (fa[:n] if fa is not None else None)
        sched = [(f_, a_, r_, min(k_, 4200 if marathon else 90)) for (f_, a_, r_, k_) in sched[:5]]
    # one round in five: a real fourth offset per opcode (IIV_OPT_FOURTH_OFFSET), against the oracle's flag -- also together
    # with the joint choice (round 6)
    fourth = bool(rng.random() < 0.2)
    enc = native.Encoder(mode, dtab[key][0], dtab[key][1], n, dm=dms[pal])
    enc.set_content_choice("split" if joint and rng.random() < 0.3 else joint)   # (both implementations of the joint choice)
    enc.set_fourth_offset(fourth)
    enc.set_diff_weights_mode(recurrence)
    enc.set_greedy_kernel(wave)
    enc.set_prefix_sort(prefix)
    seeds = [(int(rng.integers(1 << 20)), int(rng.integers(1 << 20))) for _ in range(n)]
    for i, (sp, sn) in enumerate(seeds):
        enc.set_state(native.STATE_RNG_PY, O.mt_seed_py(sp).state_words(), i)
        enc.set_state(native.STATE_RNG_NP, O.mt_seed_np(sn).state_words(), i)
    # one round in four (of the short ones): every stream its own schedule (iiv_encode_streams: streams on different banks
    # in the same launch round, streams that idle while others still work, a stream with nothing to do at all)
    per_stream = bool(not marathon and not joint and rng.random() < 0.25)
    if per_stream:
        scheds = [random_schedule(1, 8) if i else [] for i in range(n)]
        scheds[int(rng.integers(1, n))] = sched
        ops_all, totals = enc.encode_streams(fm.cuda(), fa.cuda() if fa is not None else None, scheds)
        ops_all = ops_all.cpu().numpy()
        got = [ops_all[i, :totals[i]] for i in range(n)]
    else:
        scheds = [sched] * n
        # one round in five: the call is made, rolled back to a snapshot (iiv_encoder_snapshot / _rollback: what the drop-in
        # Video's speculation rests on) and made again -- state, generators and both RNG streams must be back where they were
        replay = bool(rng.random() < 0.2)
        if replay:
            enc.snapshot()
            first = enc.encode(fm.cuda(), fa.cuda() if fa is not None else None, sched).cpu().numpy()
            enc.check()
            enc.rollback()
        got = enc.encode(fm.cuda(), fa.cuda() if fa is not None else None, sched).cpu().numpy()
        if replay:
            assert (first == got).all(), ("replay after rollback", rnd)
    enc.check()
    for i in range(n):
        v = O.Video(mode, otab[key], seed_py=seeds[i][0], seed_np=seeds[i][1])
        v.set_joint(joint)
        v.set_fourth_offset(fourth)
        exp = []
        for (fr, a, restart, k) in scheds[i]:
            if restart:
                v.encode_frame(fm[i, fr].numpy(), fa[i, fr].numpy() if fa is not None else None, a)
            if k:
                exp.append(v.next(k))
        exp = np.concatenate(exp) if exp else np.zeros((0, 6), np.uint8)
        assert (got[i] == exp).all(), ("opcodes", rnd, i, mode, pal, kind, wave, recurrence, prefix, sched)
        assert (enc.get_state(native.STATE_UP_MAIN, i) == v.update_priority(0)).all(), ("up", rnd, i)
        assert (enc.get_state(native.STATE_MEM_MAIN, i) == v.memory(0)).all(), ("mem", rnd, i)
        if mode == 1:
            assert (enc.get_state(native.STATE_UP_AUX, i) == v.update_priority(1)).all(), ("up aux", rnd, i)
        cnt = enc.get_state(native.STATE_COUNTERS, i)
        assert (int(cnt[0]), int(cnt[1])) == v.draws(), ("draws", rnd, i)
        total_ops += exp.shape[0]
    enc.close()
    print("round %2d ok: mode=%s pal=%d %s wave=%s rec=%s prefix=%d joint=%d fourth=%d%s segs=%s" % (
        rnd, "DHGR" if mode else "HGR", pal, kind, wave, recurrence, prefix, joint, fourth, " marathon" if marathon else " per-stream schedules" if per_stream else "", [s[3] for s in sched]), flush=True)
print("fuzz parity: %d rounds, %d opcodes compared, all equal (%.0f s)" % (rounds, total_ops, time.time() - t_start))
