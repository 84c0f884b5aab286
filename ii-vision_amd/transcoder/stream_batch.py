"""Many independent video streams encoded together on one GPU.

The reference encodes one video per process; its per-frame driver is
movie.Movie.encode / emit_stream (transcoder/movie.py:56-150), which pulls exactly
one opcode per audio sample, starts a fresh encode_frame() generator at every new
video frame, and (DHGR) flips between the MAIN and AUX banks whenever the output
byte stream reaches the end of a 2 KiB socket frame.  `MovieClock` restates that
control flow without audio as a list of segments, and `StreamBatch` runs the same
schedule for S independent streams -- one workgroup per stream per launch -- through
iiv_encode() (include/iivision.h).  Streams never exchange data.
"""

import numpy as np

import _iiv_native as native

TICK_OPCODE_BYTES = 7   # addr_hi, addr_lo, content, 4 offsets (opcodes.py tick opcodes)
HEADER_BYTES = 7        # opcodes.Header
ACK_BYTES = 4           # opcodes.Ack
SOCKET_FRAME = 2048


class MovieClock:
    """movie.Movie.encode/emit_stream pacing (movie.py:56-150, video.py:64-70).

    One tick = one audio sample = one opcode (movie.py:67-111).  The clock is stateful:
    consecutive calls continue one movie, and a generator that is still live when a call
    returns (same target, same bank, no new encoded frame, no bank flip) is *continued*
    by the next call (restart == 0), exactly as the reference keeps pulling from
    `op_seq` -- with every_n_video_frames > 1 a call boundary can fall on a frame that
    is not encoded (movie.py:76-80)."""

    def __init__(self, dhgr, ticks_per_second=14700.0, input_frame_rate=30.0, every_n_video_frames=1):
        self.dhgr = bool(dhgr)
        self.ticks_per_frame = float(ticks_per_second) / float(input_frame_rate)
        self.every_n = int(every_n_video_frames)
        self.ticks = 0
        self.frame_number = 0            # Video.frame_number
        self.stream_pos = HEADER_BYTES   # Movie.stream_pos after the header opcode
        self.aux_bank = False            # Movie.aux_memory_bank
        self._last_bank = False          # movie.py:66 last_memory_bank
        self._target = None              # frame index being encoded
        self._live = None                # (target, bank) of the generator op_seq refers to

    def segments(self, n_video_frames=None, max_ticks=None):
        """Segments (frame, is_aux, restart, n_ops) covering the next n_video_frames
        input frames (frame f is encoded if f % every_n == 0, movie.py:76-80) and / or at
        most max_ticks further audio samples, whichever ends first.  Stopping in front of
        frame N is what `next(video_frames)` raising StopIteration does for an N-frame
        clip (movie.py:71-74): the tick that would have started frame N emits nothing."""
        if n_video_frames is None and max_ticks is None:
            raise ValueError("segments() needs a frame count or a tick count")
        segs = []
        cur = None
        end_frame = None if n_video_frames is None else self.frame_number + int(n_video_frames)
        end_ticks = None if max_ticks is None else self.ticks + int(max_ticks)
        tpf = self.ticks_per_frame
        while True:
            if end_ticks is not None and self.ticks >= end_ticks:
                break
            # would the next tick start frame `end_frame`?  then stop before it
            nxt = self.ticks + 1
            if end_frame is not None and nxt >= tpf * self.frame_number and self.frame_number >= end_frame:
                break
            self.ticks = nxt
            restart = False
            if self.ticks >= tpf * self.frame_number:                    # Video.tick
                self.frame_number += 1
                if (self.frame_number - 1) % self.every_n == 0:
                    self._target = self.frame_number - 1
                    restart = True                                       # movie.py:94
            if self.aux_bank != self._last_bank:                         # movie.py:98-102
                self._last_bank = self.aux_bank
                restart = True
            if self._target is None:
                continue
            gen = (self._target, int(self.aux_bank))
            if restart or self._live != gen:
                cur = [gen[0], gen[1], 1, 0]
                segs.append(cur)
                self._live = gen
            elif cur is None:
                # first opcode of this call, pulled from the generator the previous call left live
                cur = [gen[0], gen[1], 0, 0]
                segs.append(cur)
            cur[3] += 1
            self.stream_pos += TICK_OPCODE_BYTES
            if self.stream_pos % SOCKET_FRAME >= SOCKET_FRAME - ACK_BYTES:  # movie.py:139-148
                if self.dhgr:
                    self.aux_bank = not self.aux_bank
                self.stream_pos += ACK_BYTES
                continue
            # ---- the ticks after this one up to the next event are all alike (same generator, one opcode
            # each): take them in one step.  Events: the tick that starts the next frame, the tick whose
            # opcode fills the socket frame (it is handled by the code above), the end of the call.
            m = int(-(-(tpf * self.frame_number) // 1)) - 1 - self.ticks      # ticks before the next frame's first
            r = self.stream_pos % SOCKET_FRAME
            m = min(m, (SOCKET_FRAME - ACK_BYTES - 1 - r) // TICK_OPCODE_BYTES)   # opcodes that still end below 2044
            if end_ticks is not None:
                m = min(m, end_ticks - self.ticks)
            if m > 0:
                self.ticks += m
                cur[3] += m
                self.stream_pos += TICK_OPCODE_BYTES * m
        return [tuple(s) for s in segs]


def merge_generators(segments):
    """[(frame, is_aux, n_ops)] per generator: restart == 0 segments folded into the
    generator they continue (what a recorder around Video.encode_frame would see)."""
    gens = []
    for (f, a, r, n) in segments:
        if r or not gens:
            gens.append([f, a, n])
        else:
            assert gens[-1][0] == f and gens[-1][1] == a
            gens[-1][2] += n
    return [tuple(g) for g in gens if g[2] > 0]


def frame_budgets(mode_is_dhgr, n_frames, **kw):
    clock = MovieClock(mode_is_dhgr, **kw)
    return clock.segments(n_frames)


def synth_frames_torch(n_streams, n_frames, dhgr, seed, coherent=False, device="cuda", keep=0.9, repeat=1):
    """SURVEY 8(d) synthetic memory maps, generated on the device: S-iid (every byte
    uniform in [0,128) DHGR / [0,256) HGR, screen holes zero) or S-coh (each byte
    keeps its previous value with probability `keep`, 0.9 in SURVEY 8d).  S-static (converging content, what
    real video with a static background does: reference README.md:39) is S-coh with keep = 0.98, optionally
    with every drawn frame shown `repeat` times in a row: the encoder's work list runs dry, it goes through its
    re-queued bag and ends out of work (video.py:124-131, 189).  Returns (main, aux) uint8
    tensors (n_streams, n_frames, 32, 256); aux is None for HGR."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    hi = 128 if dhgr else 256
    holes = (torch.arange(256, device=device) & 127) >= 120
    out = []
    for _ in range(2 if dhgr else 1):
        t = torch.randint(0, hi, (n_streams, n_frames, 32, 256), dtype=torch.uint8, device=device, generator=g)
        if coherent:
            for f in range(1, n_frames):
                if f % repeat:
                    t[:, f] = t[:, f - 1]
                    continue
                k = torch.rand((n_streams, 32, 256), device=device, generator=g) < keep
                t[:, f] = torch.where(k, t[:, f - 1], t[:, f])
        t[..., holes] = 0
        out.append(t)
    return out[0], (out[1] if dhgr else None)


def synth_frames_img(n_streams, n_frames, dhgr, seed, device="cuda"):
    """SURVEY 8(d) S-img: 1-bit dot fields -- moving grey-ramp bars rendered with a 4x4
    ordered dither -- packed 7 dots per byte through X_Y_TO_PAGE / X_Y_TO_OFFSET
    (DHGR: aux, main bytes alternate along a row).  Image-like input: large
    coherent areas, many equal windows, many delta ties.  Every stream gets its own
    bar period / speed / slope / phase.  Returns (main, aux) as synth_frames_torch."""
    import torch
    import screen
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    S = n_streams
    W = 560 if dhgr else 280

    def par(lo, hi):
        return torch.randint(lo, hi, (S, 1, 1), generator=g).to(device)

    period, speed, slope, phase = par(24, 120), par(1, 6), par(-2, 3), par(0, 120)
    y = torch.arange(192, device=device).view(1, 192, 1)
    x = torch.arange(W, device=device).view(1, 1, W)
    bayer = torch.tensor([[0, 8, 2, 10], [12, 4, 14, 6], [3, 11, 1, 9], [15, 7, 13, 5]], device=device)
    thr = bayer[y % 4, x % 4]                                       # (1,192,W)
    lin = torch.from_numpy(screen.X_Y_TO_PAGE.astype(np.int64) * 256 + screen.X_Y_TO_OFFSET.astype(np.int64))
    lin = lin.reshape(-1).to(device)                                # (192*40,) page*256+offset, row-major (y, x)
    weights = (1 << torch.arange(7, device=device)).view(1, 1, 1, 7)
    nb = 2 if dhgr else 1
    out = [torch.zeros((S, n_frames, 8192), dtype=torch.uint8, device=device) for _ in range(nb)]
    for f in range(n_frames):
        t = (x + slope * y + speed * f + phase) % period            # (S,192,W)
        dots = ((t * 17) // period) > thr
        by = (dots.view(S, 192, W // 7, 7) * weights).sum(-1).to(torch.uint8)   # (S,192,W/7)
        if dhgr:
            out[1][:, f, lin] = by[:, :, 0::2].reshape(S, -1)      # aux bytes come first in dot order
            out[0][:, f, lin] = by[:, :, 1::2].reshape(S, -1)
        else:
            out[0][:, f, lin] = by.reshape(S, -1)
    out = [o.view(S, n_frames, 32, 256) for o in out]
    return out[0], (out[1] if dhgr else None)


def synth_rgb_torch(n_clips, n_frames, seed, device="cuda"):
    """Synthetic RGB video for the frame-ingest path (f3: iiv_frames_to_memory_maps): (n_clips, n_frames, 192, 280, 3) uint8,
    picture-like -- per clip three drifting low-frequency colour fields, a moving hard-edged bar and a little noise, so that
    dithering has gradients to work on and consecutive frames resemble each other.  Generated on the device."""
    import torch
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))

    def par(lo, hi, shape=(n_clips, 1, 1, 3)):
        return (torch.rand(shape, generator=g) * (hi - lo) + lo).to(device)

    fx, fy, ft, ph = par(0.004, 0.03), par(0.004, 0.04), par(0.02, 0.12), par(0.0, 6.28)
    bar_speed, bar_w, bar_pos = par(0.5, 4.0, (n_clips, 1, 1, 1)), par(10.0, 60.0, (n_clips, 1, 1, 1)), par(0.0, 280.0, (n_clips, 1, 1, 1))
    bar_col = par(0.0, 255.0, (n_clips, 1, 1, 3))
    y = torch.arange(192, device=device, dtype=torch.float32).view(1, 192, 1, 1)
    x = torch.arange(280, device=device, dtype=torch.float32).view(1, 1, 280, 1)
    gd = torch.Generator(device=device)
    gd.manual_seed(int(seed) + 1)
    out = torch.empty((n_clips, n_frames, 192, 280, 3), dtype=torch.uint8, device=device)
    for f in range(n_frames):
        v = 127.5 + 127.5 * torch.sin(fx * x + fy * y + ft * f + ph)                       # (n_clips, 192, 280, 3)
        bx = (bar_pos + bar_speed * f) % 280.0
        in_bar = ((x - bx) % 280.0) < bar_w
        v = torch.where(in_bar, bar_col.expand_as(v), v)
        v = v + (torch.rand(v.shape, device=device, generator=gd) - 0.5) * 6.0
        out[:, f] = v.clamp_(0, 255).to(torch.uint8)
    return out


class StreamBatch:
    """S independent video.Video encoders advanced in lock step on one GPU."""

    def __init__(self, mode, table, store_table, n_streams, seeds=None, dm=None, joint_content=False, fourth_offset=False,
                 use_torch_ops=True, **clock_kw):
        """joint_content=True: the content byte of every step is chosen jointly with its extra
        offsets (reference README.md:212-215, include/iivision.h IIV_CONTENT_JOINT) -- better
        pictures per opcode, NOT the reference's opcode stream.  fourth_offset=True: every opcode stores at
        up to four distinct offsets instead of three and a repeat (IIV_OPT_FOURTH_OFFSET) -- likewise."""
        self.mode = mode
        self.n_streams = int(n_streams)
        # the launches go through torch.ops.iivision.encode (torch_ops.py: the C ABI registered as PyTorch custom operators);
        # use_torch_ops=False calls the same C entry point through ctypes directly -- same kernels, same bytes (tests)
        self.use_torch_ops = bool(use_torch_ops)
        self.enc = native.Encoder(mode, table, store_table, self.n_streams, dm=dm)
        if joint_content:
            self.enc.set_content_choice(joint_content)   # (True, or "split" for the second implementation)
        if fourth_offset:
            self.enc.set_fourth_offset(True)
        self.clock = MovieClock(mode == native.DHGR, **clock_kw)
        if seeds is not None:
            self.seed(seeds)

    def seed(self, seeds):
        """seeds[i] = (random.seed value, np.random.seed value) of stream i."""
        import random
        keep_py, keep_np = random.getstate(), np.random.get_state()
        n = len(seeds)
        py = np.empty((n, 625), dtype=np.uint32)
        nps = np.empty((n, 625), dtype=np.uint32)
        try:
            for i, (sp, sn) in enumerate(seeds):
                random.seed(int(sp))
                np.random.seed(int(sn))
                st = np.random.get_state()
                py[i] = random.getstate()[1]
                nps[i, :624] = st[1]
                nps[i, 624] = st[2]
        finally:
            random.setstate(keep_py)
            np.random.set_state(keep_np)
        # one upload per RNG stream for the whole batch
        self.enc.set_state_all(native.STATE_RNG_PY, py)
        self.enc.set_state_all(native.STATE_RNG_NP, nps)

    def encode_frames(self, frames_main, frames_aux, n_video_frames, ops_out=None, max_ticks=None, loop=False):
        """Advance every stream by n_video_frames input frames (targets are
        frames_*[:, clock.frame_number ...]) or max_ticks audio samples, whichever ends
        first; returns (ops tensor, segments).  loop=True: the clip repeats -- frame f of the movie is
        frames_*[:, f % clip length] (the returned segments keep the movie's frame numbers)."""
        segs = self.clock.segments(n_video_frames, max_ticks=max_ticks)
        n = int(frames_main.shape[1])
        plan = [(f % n, a, r, k) for (f, a, r, k) in segs] if loop else segs
        if self.use_torch_ops:
            import torch_ops
            ops = torch_ops.encode_via_op(self.enc, frames_main, frames_aux, plan, ops_out)
        else:
            ops = self.enc.encode(frames_main, frames_aux, plan, ops_out)
        return ops, segs

    def close(self):
        self.enc.close()
