"""Where a drop-in frame's wall time goes (tools/dropin_profile.py's tick-paced loop, unprofiled): perf_counter stamps around
the pieces of a generator start and the hand-out.   python tools/dropin_timeline.py   (needs the GPU)"""
import contextlib, io, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ii-vision_amd", "transcoder"))
import numpy as np
import palette, screen, stream_batch, video, video_mode

n_frames = 40
pal = palette.Palette.NTSC
fm, fa = stream_batch.synth_frames_torch(1, n_frames, True, seed=3, device="cpu")
acc = {}


def timed(obj, name, label=None):
    f = getattr(obj, name)
    label = label or name

    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            d = acc.setdefault(label, [0, 0.0])
            d[0] += 1
            d[1] += time.perf_counter() - t
    setattr(obj, name, g)


class FG:
    input_frame_rate = 30


def run(instrument):
    random.seed(1)
    np.random.seed(1)
    v = video.Video(FG(), ticks_per_second=14700., palette=pal, mode=video_mode.VideoMode.DHGR)
    if instrument:
        for name in ("_sync_brief", "_settle", "_look_ahead", "_launch_live", "_adopt", "_live_take", "_paced_chunk", "_global_rng_moved", "_set_global_rng"):
            timed(v, name)
        for name in ("encode_live", "snapshot", "get_video_brief_async", "rollback", "check"):
            timed(v._enc, name, "enc." + name)
    tgts = {}

    def target_of(fr):
        if fr not in tgts:
            tgts[fr] = screen.DHGRBitmap(main_memory=screen.MemoryMap(1, fm[0, fr].numpy().copy()),
                                         aux_memory=screen.MemoryMap(1, fa[0, fr].numpy().copy()), palette=pal)
        return tgts[fr]
    for fr in range(n_frames):
        target_of(fr)     # (the caller's own cost of making a Bitmap is not what is looked at here)
    t_next = 0.0
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        ticks, stream_pos, aux, last_bank, op_seq, target = 0, 7, False, False, None, None
        while True:
            ticks += 1
            if v.tick(ticks):
                if v.frame_number - 1 >= n_frames:
                    break
                target = target_of(v.frame_number - 1)
                op_seq = v.encode_frame(target, is_aux=aux)
                v.out_of_work = {True: False, False: False}
            if aux != last_bank:
                last_bank = aux
                op_seq = v.encode_frame(target, is_aux=aux)
            next(op_seq)
            stream_pos += 7
            if stream_pos % 2048 >= 2044:
                aux = not aux
                stream_pos += 4
    dt = time.perf_counter() - t0
    return dt, v


run(False)
dt, v = run(False)
print("plain: %.0f us per frame (%.0f frames/s); live_stats per frame: %s" % (1e6 * dt / n_frames, n_frames / dt,
      {k: (round(1e6 * x / n_frames, 1) if isinstance(x, float) else round(x / n_frames, 2)) for k, x in v.live_stats.items()}))
acc.clear()
dt, v = run(True)
print("instrumented: %.0f us per frame" % (1e6 * dt / n_frames))
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("  %-26s %6.2f calls per frame  %7.1f us per frame  %6.1f us per call" % (k, n / n_frames, 1e6 * t / n_frames, 1e6 * t / n))
