"""Diagnostic: cProfile of the drop-in Video path (no budget hint), DHGR, 20 frames."""
import os, sys, time, random, io, contextlib, cProfile, pstats
sys.path.insert(0, os.path.join(os.getcwd(), 'ii-vision_amd', 'transcoder'))
import numpy as np
import screen, video, video_mode, palette, stream_batch
class FG: input_frame_rate = 30
fm, fa = stream_batch.synth_frames_torch(1, 20, True, seed=3, device="cpu")
def run(spec, budget):
    random.seed(1); np.random.seed(1)
    v = video.Video(FG(), ticks_per_second=14700., mode=video_mode.VideoMode.DHGR, palette=palette.Palette.NTSC)
    v.SPECULATE = spec
    segs = stream_batch.MovieClock(True).segments(20)
    tgts = {}
    t0 = time.perf_counter(); n = 0
    with contextlib.redirect_stdout(io.StringIO()):
        for (fr, ia, _, k) in segs:
            if fr not in tgts:
                tgts[fr] = screen.DHGRBitmap(main_memory=screen.MemoryMap(1, fm[0, fr].numpy().copy()),
                                             aux_memory=screen.MemoryMap(1, fa[0, fr].numpy().copy()), palette=palette.Palette.NTSC)
            gen = v.encode_frame(tgts[fr], is_aux=bool(ia), budget=k if budget else None)
            for _ in range(k):
                next(gen); n += 1
    return 20 / (time.perf_counter() - t0)
run(64, False)
for spec, budget in ((64, False), (256, False), (512, False), (0, True)):
    print("SPECULATE=%d budget=%s: %.1f frames/s" % (spec, budget, run(spec, budget)), flush=True)
pr = cProfile.Profile(); pr.enable(); run(64, False); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:5000])
