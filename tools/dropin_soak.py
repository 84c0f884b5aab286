"""Soak of the drop-in video.Video (live hand-over + look-ahead) against the oracle: N frames driven statement by statement as
movie.Movie.encode drives it (tick() per audio sample, a generator per frame and bank flip, one next() per opcode), a 64-frame
S-coh clip looping, every opcode compared.  N = 26000 DHGR frames is 69 000 launches: the live queue's 16-bit tag starts over
on the way.   python tools/dropin_soak.py [DHGR|HGR] [frames]     (needs the GPU; the oracle runs on one host core)"""
import contextlib, io, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ii-vision_amd", "transcoder"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle as O
import palette, screen, stream_batch, video, video_mode

dhgr = "HGR" not in sys.argv[1:]
n_frames = next((int(a) for a in sys.argv[1:] if a.isdigit()), 2000)
CLIP = 64
O.build()
mode = 1 if dhgr else 0
pal = palette.Palette.NTSC
fm, fa = stream_batch.synth_frames_torch(1, CLIP, dhgr, seed=9, coherent=True, device="cpu")
fm = fm[0].numpy()
fa = fa[0].numpy() if dhgr else None


class FG:
    input_frame_rate = 30


random.seed(3)
np.random.seed(4)
v = video.Video(FG(), ticks_per_second=14700., palette=pal, mode=video_mode.VideoMode.DHGR if dhgr else video_mode.VideoMode.HGR)
tgts = [screen.DHGRBitmap(main_memory=screen.MemoryMap(1, fm[f].copy()), aux_memory=screen.MemoryMap(1, fa[f].copy()), palette=pal) if dhgr
        else screen.HGRBitmap(main_memory=screen.MemoryMap(1, fm[f].copy()), palette=pal) for f in range(CLIP)]
got = np.empty((n_frames * 491 + 16, 6), dtype=np.uint8)
n_got = 0
t0 = time.perf_counter()
with contextlib.redirect_stdout(io.StringIO()):
    ticks, stream_pos, aux, last_bank, op_seq, target = 0, 7, False, False, None, None
    while True:
        ticks += 1
        if v.tick(ticks):
            if v.frame_number - 1 >= n_frames:
                break
            target = tgts[(v.frame_number - 1) % CLIP]
            op_seq = v.encode_frame(target, is_aux=aux)
            v.out_of_work = {True: False, False: False}
        if aux != last_bank:
            last_bank = aux
            op_seq = v.encode_frame(target, is_aux=aux)
        page, content, offsets = next(op_seq)
        row = got[n_got]
        row[0], row[1] = page, content
        row[2:] = offsets
        n_got += 1
        stream_pos += 7
        if stream_pos % 2048 >= 2044:
            if dhgr:
                aux = not aux
            stream_pos += 4
dt = time.perf_counter() - t0
op_seq = None
print("%s: %d frames, %d opcodes through video.Video in %.1f s (%.0f frames/s with the per-opcode copy into the log); live: %s; look-ahead: %s; tag epoch %d"
      % ("DHGR" if dhgr else "HGR", n_frames, n_got, dt, n_frames / dt, v.live_stats, v.lookahead_stats, v._live_epoch))
sys.stdout.flush()
t0 = time.perf_counter()
dm = O.cie2000_matrix(O.PALETTE_RGB[5])[1]
ov = O.Video(mode, O.build_table(mode, dm, symmetric=True), seed_py=3, seed_np=4)
pos, prev, bad = 0, None, None
for (f, a, _, k) in stream_batch.MovieClock(dhgr).segments(n_frames):
    if f != prev:
        ov.reset_out_of_work()
        prev = f
    ov.encode_frame(fm[f % CLIP], fa[f % CLIP] if dhgr else None, int(a))
    want = ov.next(k)
    if not (got[pos:pos + k] == want).all():
        bad = (f, a, pos + int(np.argmax((got[pos:pos + k] != want).any(axis=1))))
        break
    pos += k
print("oracle: %d opcodes in %.0f s: %s" % (pos, time.perf_counter() - t0, "ALL EQUAL" if bad is None and pos == n_got else "MISMATCH at frame %s bank %s opcode %s (of %d)" % (bad + (n_got,) if bad else ("-", "-", pos, n_got))))
assert bad is None and pos == n_got
