"""Where the drop-in video.Video path (one next() per opcode from Python) spends its time: cProfile of 20 Movie-paced
frames, with and without encode_frame(budget=K), and ("paced") driven statement by statement as movie.Movie.encode does
(tick() per audio sample: the form bench.py's `dropin` leg times).   python tools/dropin_profile.py [DHGR|HGR]"""
import contextlib, cProfile, io, os, pstats, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ii-vision_amd", "transcoder"))
import numpy as np
import palette, screen, stream_batch, video, video_mode

dhgr = (sys.argv[1] if len(sys.argv) > 1 else "DHGR") == "DHGR"
n_frames = 20
pal = palette.Palette.NTSC
fm, fa = stream_batch.synth_frames_torch(1, n_frames, dhgr, seed=3, device="cpu")


class FrameGrabber:
    input_frame_rate = 30


def run(budget):
    random.seed(1)
    np.random.seed(1)
    v = video.Video(FrameGrabber(), ticks_per_second=14700., palette=pal,
                    mode=video_mode.VideoMode.DHGR if dhgr else video_mode.VideoMode.HGR)
    segs = stream_batch.MovieClock(dhgr).segments(n_frames)
    tgts = {}

    def target_of(fr):
        if fr not in tgts:
            main = screen.MemoryMap(1, fm[0, fr].numpy().copy())
            tgts[fr] = (screen.DHGRBitmap(main_memory=main, aux_memory=screen.MemoryMap(1, fa[0, fr].numpy().copy()), palette=pal)
                        if dhgr else screen.HGRBitmap(main_memory=main, palette=pal))
        return tgts[fr]

    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        if budget == "paced":
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
                    if dhgr:
                        aux = not aux
                    stream_pos += 4
            dt = time.perf_counter() - t0
            ls = getattr(v, "live_stats", None)
            if ls:
                print("   live hand-over: %s; per frame: %.0f us in all, %.0f us waiting for the device" % (ls, 1e6 * dt / n_frames, 1e6 * ls["wait_s"] / n_frames), file=sys.__stdout__)
            return n_frames / dt
        for (fr, ia, _, k) in segs:
            gen = v.encode_frame(target_of(fr), is_aux=bool(ia), **({"budget": k} if budget else {}))
            for _ in range(k):
                next(gen)
    return n_frames / (time.perf_counter() - t0)


run(False)
for budget in ("paced", False, True):
    print("budget=%s: %.0f frames/s" % (budget, run(budget)))
    pr = cProfile.Profile()
    pr.enable()
    run(budget)
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(26)
    print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:6000])
