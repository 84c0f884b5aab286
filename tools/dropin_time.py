import os, sys, time, random, io, contextlib
sys.path.insert(0, os.path.join(os.getcwd(), 'ii-vision_amd', 'transcoder'))
import numpy as np
import screen, video, video_mode, palette, stream_batch
class FG: input_frame_rate = 30
fm, fa = stream_batch.synth_frames_torch(1, 20, True, seed=3, device="cpu")
clock = stream_batch.MovieClock(True)
for spec, budget in ((0, False), (64, False), (256, False), (0, True)):
    random.seed(1); np.random.seed(1)
    v = video.Video(FG(), ticks_per_second=14700., mode=video_mode.VideoMode.DHGR, palette=palette.Palette.NTSC)
    v.SPECULATE = spec
    clk = stream_batch.MovieClock(True)
    segs = clk.segments(20)
    t0 = time.perf_counter(); n = 0
    with contextlib.redirect_stdout(io.StringIO()):
        for (fr, ia, _, k) in segs:
            tgt = screen.DHGRBitmap(main_memory=screen.MemoryMap(1, fm[0, fr].numpy().copy()),
                                    aux_memory=screen.MemoryMap(1, fa[0, fr].numpy().copy()), palette=palette.Palette.NTSC)
            gen = v.encode_frame(tgt, is_aux=bool(ia), budget=k if budget else None)
            for _ in range(k):
                next(gen); n += 1
    dt = time.perf_counter() - t0
    print("SPECULATE=%d budget=%s: %d opcodes in %.2f s = %.1f frames/s" % (spec, budget, n, dt, 20 / dt))
