#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing and running the REFERENCE.

Runs only in the build container (it needs /root/reference, which does not exist
on the GPU box).  Nothing here is used at test time; the tests read the .npz
fixtures this script wrote.  Fixtures hold inputs and expected outputs only.

Recipe:
  * NumPy-2 shim: np.bool8 / np.int were removed; the reference uses them at
    import (screen.py:42, audio.py:97).
  * stub modules for packages that are absent here and only hold data / are
    never called on this path: colormath.color_objects.sRGBColor (palette.py:6-15),
    skvideo.io (frame_grabber.py:10).
  * cwd = scratch dir with player -> /root/reference/player (opcodes.py:173) and
    transcoder/data/*.npz (screen.py:348).
  * the .npz tables the reference loads are produced by the oracle's table
    builder in the reference's own on-disk format (lower triangle, key
    'edit_distance'); make_data_tables.py itself cannot run here because
    colormath / weighted_levenshtein are absent, so table VALUES stay unpinned
    while everything downstream of the tables is pinned.

Usage: python tests/golden/make_golden.py [--scratch /tmp/iiv_ref]
"""

import argparse
import hashlib
import os
import random
import sys
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def setup_reference(scratch):
    os.makedirs(os.path.join(scratch, "transcoder", "data"), exist_ok=True)
    link = os.path.join(scratch, "player")
    if not os.path.exists(link):
        os.symlink(os.path.join(REF, "player"), link)
    np.bool8 = np.bool_
    np.int = int

    colormath = types.ModuleType("colormath")
    co = types.ModuleType("colormath.color_objects")

    class sRGBColor:
        def __init__(self, r, g, b, is_upscaled=False):
            self.rgb = (r, g, b)
            self.is_upscaled = is_upscaled

    class LabColor:
        pass

    co.sRGBColor = sRGBColor
    co.LabColor = LabColor
    colormath.color_objects = co
    sys.modules["colormath"] = colormath
    sys.modules["colormath.color_objects"] = co
    sk = types.ModuleType("skvideo")
    skio = types.ModuleType("skvideo.io")
    sk.io = skio
    sys.modules["skvideo"] = sk
    sys.modules["skvideo.io"] = skio
    # movie.py imports audio.py, which imports these at module level; neither is called
    # by the byte-emission path exercised for g6
    for name in ("audioread", "librosa"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.path.insert(0, os.path.join(REF, "transcoder"))
    os.chdir(scratch)


def write_reference_tables(scratch, O):
    """Oracle-built tables in the reference's file format, for the reference to load."""
    out = {}
    for pal in (5, 0):
        _, dm = O.cie2000_matrix(O.PALETTE_RGB[pal])
        for mode, name in ((O.HGR, "HGR"), (O.DHGR, "DHGR")):
            path = os.path.join(scratch, "transcoder", "data",
                                "%s_palette_%d_edit_distance.npz" % (name, pal))
            if not os.path.exists(path):
                t = time.time()
                tab = O.build_table(mode, dm, symmetric=False)
                np.savez(path, edit_distance=tab)
                print("built %s in %.1fs" % (path, time.time() - t))
            out[(name, pal)] = path
    return out


def synth_frames(mode_name, n_frames, seed, coherent=False):
    """SURVEY.md 8(d) synthetic memory maps: S-iid / S-coh."""
    import screen
    rng = np.random.default_rng(seed)
    hi = 128 if mode_name == "DHGR" else 256
    banks = 2 if mode_name == "DHGR" else 1
    frames = np.zeros((n_frames, banks, 32, 256), dtype=np.uint8)
    prev = None
    for f in range(n_frames):
        cur = []
        for b in range(banks):
            new = rng.integers(0, hi, (32, 256), dtype=np.uint8)
            if coherent and prev is not None:
                keep = rng.random((32, 256)) < 0.9
                new = np.where(keep, prev[b], new)
            new[screen.SCREEN_HOLES] = 0
            cur.append(new)
        prev = cur
        for b in range(banks):
            frames[f, b] = cur[b]
    return frames


def run_reference(mode_name, palette_id, frames, schedule, seed_py, seed_np):
    """Drive reference video.Video with a list of (frame_idx, is_aux, n_ops) segments.

    Each segment creates a fresh generator (as movie.py:94,101 do) and pulls n_ops.
    """
    import frame_grabber
    import palette
    import screen
    import video
    import video_mode

    mode = video_mode.VideoMode[mode_name]
    pal = palette.Palette(palette_id)
    random.seed(seed_py)
    np.random.seed(seed_np)
    fg = frame_grabber.FrameGrabber(mode)
    v = video.Video(fg, ticks_per_second=14700., mode=mode, palette=pal)
    ops = []
    import io
    import contextlib
    for (fi, is_aux, n_ops) in schedule:
        main = screen.MemoryMap(screen_page=1, page_offset=frames[fi, 0].copy())
        if mode_name == "DHGR":
            aux = screen.MemoryMap(screen_page=1, page_offset=frames[fi, 1].copy())
            tgt = screen.DHGRBitmap(main_memory=main, aux_memory=aux, palette=pal)
        else:
            tgt = screen.HGRBitmap(main_memory=main, palette=pal)
        gen = v.encode_frame(tgt, is_aux=bool(is_aux))
        with contextlib.redirect_stdout(io.StringIO()):
            for _ in range(n_ops):
                page, content, offsets = next(gen)
                ops.append([page, int(content)] + [int(o) for o in offsets])
    ops = np.array(ops, dtype=np.uint8).reshape(-1, 6)
    state = dict(
        ops=ops,
        mem_main=v.memory_map.page_offset.copy(),
        up_main=v.update_priority.copy(),
        packed=v.pixelmap.packed.copy(),
        out_of_work=np.array([v.out_of_work[False], v.out_of_work[True]], dtype=np.uint8),
    )
    if mode_name == "DHGR":
        state["mem_aux"] = v.aux_memory_map.page_offset.copy()
        state["up_aux"] = v.aux_update_priority.copy()
    # RNG positions after the run, as "next 4 outputs" of each stream
    state["py_next"] = np.array([random.getrandbits(8) for _ in range(4)], dtype=np.uint8)
    state["np_next"] = np.random.randint(0, 256, size=4).astype(np.uint8)
    return state


def movie_schedule(mode_name, n_frames, ops_per_frame=490):
    """movie.py:56-150 control flow without audio: frame 0 gets
    ops_per_frame-1 opcodes; DHGR flips banks when the 7-byte-per-opcode stream
    position reaches 2044 mod 2048."""
    sched = []
    pos = 7  # header
    is_aux = 0
    ticks = 0
    frame_number = 0
    ticks_per_frame = float(ops_per_frame)
    cur = None
    total_ticks = n_frames * ops_per_frame - 1
    fi = -1
    for _ in range(total_ticks):
        ticks += 1
        new_frame = False
        if ticks >= ticks_per_frame * frame_number:
            frame_number += 1
            fi += 1
            new_frame = True
        if new_frame or cur is None or cur[1] != is_aux:
            cur = [fi, is_aux, 0]
            sched.append(cur)
        cur[2] += 1
        pos += 7
        if pos % 2048 >= 2044:
            pos += 4
            if mode_name == "DHGR":
                is_aux ^= 1
    return [tuple(s) for s in sched if s[2] > 0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scratch", default="/tmp/iiv_ref")
    ap.add_argument("--a2m-only", action="store_true")
    ap.add_argument("--movie-only", action="store_true")
    ap.add_argument("--fourth-only", action="store_true")
    args = ap.parse_args()

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O

    setup_reference(args.scratch)
    write_reference_tables(args.scratch, O)

    import colours
    import screen

    # ---- G2: to_dots + colour-pixel strings for every masked value ----------
    g2 = {}
    for name, cls, ncol in (("HGR", screen.HGRBitmap, colours.HGRColours),
                            ("DHGR", screen.DHGRBitmap, colours.DHGRColours)):
        bits = int(cls.MASKED_BITS)
        nd = int(cls.MASKED_DOTS)
        dots = np.zeros((len(cls.PHASES), 1 << bits), dtype=np.uint32)
        pix = np.zeros((len(cls.PHASES), 1 << bits, nd), dtype=np.uint8)
        for o, ph in enumerate(cls.PHASES):
            for i in range(1 << bits):
                d = cls.to_dots(i, byte_offset=o)
                dots[o, i] = d
                pix[o, i] = colours.dots_to_nominal_colour_pixel_values(nd, d, ncol, init_phase=ph)
        g2[name + "_dots"] = dots
        g2[name + "_pixels"] = pix
    np.savez_compressed(os.path.join(HERE, "g2_dots_pixels.npz"), **g2)
    print("g2 written")

    # ---- geometry ------------------------------------------------------------
    np.savez_compressed(
        os.path.join(HERE, "g0_geometry.npz"),
        screen_holes=screen.SCREEN_HOLES.astype(np.uint8),
        x_y_to_page=screen.X_Y_TO_PAGE, x_y_to_offset=screen.X_Y_TO_OFFSET,
        page_offset_to_x=screen.PAGE_OFFSET_TO_X, page_offset_to_y=screen.PAGE_OFFSET_TO_Y)

    # ---- G4: pack / apply / diff_weights / compute_delta_page on random state -
    import palette
    g4 = {}
    rng = np.random.default_rng(1234)
    for name in ("HGR", "DHGR"):
        pal = palette.Palette.NTSC
        hi = 128 if name == "DHGR" else 256

        def mk():
            a = rng.integers(0, hi, (32, 256), dtype=np.uint8)
            return a

        src_main, src_aux, tgt_main, tgt_aux = mk(), mk(), mk(), mk()
        if name == "DHGR":
            src = screen.DHGRBitmap(pal, screen.MemoryMap(1, src_main.copy()), screen.MemoryMap(1, src_aux.copy()))
            tgt = screen.DHGRBitmap(pal, screen.MemoryMap(1, tgt_main.copy()), screen.MemoryMap(1, tgt_aux.copy()))
            banks = (False, True)
        else:
            src = screen.HGRBitmap(pal, screen.MemoryMap(1, src_main.copy()))
            tgt = screen.HGRBitmap(pal, screen.MemoryMap(1, tgt_main.copy()))
            banks = (False,)
        g4[name + "_src_main"], g4[name + "_src_aux"] = src_main, src_aux
        g4[name + "_tgt_main"], g4[name + "_tgt_aux"] = tgt_main, tgt_aux
        g4[name + "_src_packed"] = src.packed.copy()
        g4[name + "_tgt_packed"] = tgt.packed.copy()
        for ia in banks:
            dw = tgt.diff_weights(src, ia)
            g4["%s_dw_%d" % (name, ia)] = dw.copy()
            pages = rng.integers(0, 32, 6)
            contents = rng.integers(0, hi, 6)
            deltas = np.zeros((6, 256), dtype=np.int32)
            for k in range(6):
                deltas[k] = tgt.compute_delta_page(int(pages[k]), int(contents[k]), dw[int(pages[k]), :], ia)
            g4["%s_delta_pages_%d" % (name, ia)] = pages.astype(np.int32)
            g4["%s_delta_contents_%d" % (name, ia)] = contents.astype(np.int32)
            g4["%s_delta_%d" % (name, ia)] = deltas
        # a random apply() sequence on the source bitmap
        n_apply = 400
        seq = np.zeros((n_apply, 4), dtype=np.int32)
        for k in range(n_apply):
            p, o = int(rng.integers(0, 32)), int(rng.integers(0, 256))
            ia = bool(rng.integers(0, 2)) if name == "DHGR" else False
            val = int(rng.integers(0, hi))
            src.apply(p, o, ia, np.uint8(val))
            seq[k] = (p, o, int(ia), val)
        g4[name + "_apply_seq"] = seq
        g4[name + "_apply_packed"] = src.packed.copy()
        g4[name + "_apply_main"] = src.main_memory.page_offset.copy()
        if name == "DHGR":
            g4[name + "_apply_aux"] = src.aux_memory.page_offset.copy()
    np.savez_compressed(os.path.join(HERE, "g4_bitmap_ops.npz"), **g4)
    print("g4 written")

    # ---- G3: seeded encode_frame runs ----------------------------------------
    g3 = {}
    cases = []
    # (tag, mode, palette, n_frames, data_seed, coherent, rng_seed, schedule or None)
    for seed in (1, 2, 3):
        cases.append(("DHGR_iid_s%d" % seed, "DHGR", 5, 3, 7, False, seed, None))
        cases.append(("HGR_iid_s%d" % seed, "HGR", 5, 3, 7, False, seed, None))
    cases.append(("DHGR_coh_s1", "DHGR", 5, 6, 7, True, 1, None))
    cases.append(("HGR_coh_s1", "HGR", 5, 6, 7, True, 1, None))
    cases.append(("DHGR_iigs_s1", "DHGR", 0, 2, 11, False, 1, None))
    cases.append(("HGR_iigs_s1", "HGR", 0, 2, 11, False, 1, None))
    # run to exhaustion -> wrapped-key phase -> padding (one frame, one generator
    # per bank, far more opcodes than there is work)
    cases.append(("HGR_exhaust", "HGR", 5, 2, 21, False, 5,
                  [(0, 0, 6500), (1, 0, 300), (1, 0, 7000)]))
    cases.append(("DHGR_exhaust", "DHGR", 5, 2, 22, False, 5,
                  [(0, 0, 5000), (0, 1, 5200), (0, 0, 900), (1, 1, 2500), (1, 0, 2500), (1, 1, 4000)]))
    # single one-op pulls (lazy generator abandoned after each op)
    cases.append(("DHGR_single_ops", "DHGR", 5, 1, 23, False, 9,
                  [(0, 0, 1), (0, 0, 1), (0, 1, 1), (0, 0, 2), (0, 1, 3)]))
    for (tag, mode_name, pal, nf, dseed, coh, rseed, sched) in cases:
        t = time.time()
        frames = synth_frames(mode_name, nf, dseed, coherent=coh)
        if sched is None:
            sched = movie_schedule(mode_name, nf)
        st = run_reference(mode_name, pal, frames, sched, rseed, rseed)
        g3[tag + "/frames"] = frames
        g3[tag + "/schedule"] = np.array(sched, dtype=np.int32)
        g3[tag + "/meta"] = np.array([0 if mode_name == "HGR" else 1, pal, rseed, rseed], dtype=np.int32)
        for k, val in st.items():
            g3[tag + "/" + k] = val
        print("%s: %d ops, sha %s (%.1fs)" % (tag, len(st["ops"]), sha(st["ops"])[:16], time.time() - t))
    np.savez_compressed(os.path.join(HERE, "g3_encode_runs.npz"), **g3)
    print("g3 written")

    # ---- G5: table hashes (values come from the ORACLE: unpinned) -------------
    g5 = {}
    for pal in (5, 0):
        f, dm = O.cie2000_matrix(O.PALETTE_RGB[pal])
        g5["dm_f_%d" % pal] = f
        g5["dm_i_%d" % pal] = dm
        for mode, name in ((O.HGR, "HGR"), (O.DHGR, "DHGR")):
            tab = np.load(os.path.join(args.scratch, "transcoder", "data",
                                       "%s_palette_%d_edit_distance.npz" % (name, pal)))["edit_distance"]
            g5["%s_%d_lower_sha256" % (name, pal)] = np.frombuffer(
                hashlib.sha256(tab.tobytes()).digest(), dtype=np.uint8)
            g5["%s_%d_lower_sum" % (name, pal)] = np.array([tab.astype(np.uint64).sum()], dtype=np.uint64)
            srng = np.random.default_rng(99)
            idx = srng.integers(0, tab.shape[1], 10000)
            o = srng.integers(0, tab.shape[0], 10000)
            g5["%s_%d_sample_o" % (name, pal)] = o.astype(np.int32)
            g5["%s_%d_sample_idx" % (name, pal)] = idx.astype(np.int64)
            g5["%s_%d_sample_val" % (name, pal)] = tab[o, idx]
    np.savez_compressed(os.path.join(HERE, "g5_tables.npz"), **g5)
    print("g5 written")
    make_a2m_golden(args.scratch)
    make_movie_golden(args.scratch)
    make_fourth_offset_golden(args.scratch)


if __name__ == "__main__" and not {"--a2m-only", "--movie-only", "--fourth-only"} & set(sys.argv):
    main()


def make_a2m_golden(scratch):
    """G6: opcode byte emission (movie.emit_stream / opcodes / machine), f2 of SURVEY 8f.
    The opcode start addresses come from the reference's player/iivision.dbg symbol table
    (data); the expected byte streams come from the reference's own Movie.emit_stream."""
    import machine
    import movie
    import opcodes
    import video_mode
    addr = np.zeros((32, 32), dtype=np.uint16)   # [tick index (tick-4)/2][page-32]
    for ti, tick in enumerate(range(4, 68, 2)):
        for page in range(32, 64):
            addr[ti, page - 32] = opcodes.TICK_OPCODES[(tick, page)]._START
    special = np.array([opcodes.Ack._START, opcodes.Terminate._START, opcodes.Nop._START], dtype=np.uint16)
    out = {"tick_addr": addr, "special_addr": special}
    rng = np.random.default_rng(77)
    for tag, mode, n_ops, max_bytes in (("HGR_a", "HGR", 700, None), ("DHGR_a", "DHGR", 1000, None),
                                        ("DHGR_b", "DHGR", 291, None), ("DHGR_c", "DHGR", 292, None),
                                        ("HGR_limit", "HGR", 900, 3000), ("DHGR_empty", "DHGR", 0, None)):
        m = movie.Movie.__new__(movie.Movie)
        m.video_mode = video_mode.VideoMode[mode]
        m.max_bytes_out = max_bytes
        m.stream_pos = 0
        m.state = machine.Machine()
        m.aux_memory_bank = False
        ticks = rng.integers(0, 32, n_ops) * 2 + 4
        ops = rng.integers(0, 256, (n_ops, 6)).astype(np.uint8)
        ops[:, 0] = rng.integers(32, 64, n_ops)

        def gen():
            yield opcodes.Header(mode=m.video_mode)
            for k in range(n_ops):
                yield opcodes.TICK_OPCODES[(int(ticks[k]), int(ops[k, 0]))](int(ops[k, 1]), tuple(int(x) for x in ops[k, 2:6]))
        stream = np.array(list(m.emit_stream(gen())), dtype=np.uint8)
        out[tag + "/ticks"] = ticks.astype(np.uint8)
        out[tag + "/ops"] = ops
        out[tag + "/stream"] = stream
        out[tag + "/meta"] = np.array([0 if mode == "HGR" else 1, -1 if max_bytes is None else max_bytes], dtype=np.int32)
        print(tag, len(stream))
    np.savez_compressed(os.path.join(HERE, "g6_a2m.npz"), **out)


if __name__ == "__main__" and "--a2m-only" in sys.argv:
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    setup_reference("/tmp/iiv_ref")
    make_a2m_golden("/tmp/iiv_ref")


def make_movie_golden(scratch):
    """G7: the reference's OWN Movie.encode / emit_stream control flow (movie.py:56-150),
    f1 of SURVEY 8f.  A movie.Movie object is made through __new__ (its __init__ opens an
    audio file and ffmpeg); `audio` yields N zero samples (tick 34 each), `frame_grabber`
    yields the synthetic memory maps, `video` is the reference's video.Video.  The byte
    stream comes out of Movie.emit_stream(Movie.encode()) exactly as main.py:67-69 pulls it.
    video.Video.encode_frame is wrapped (not changed) by a recorder that logs, per call,
    (index of the target frame, is_aux, opcodes pulled from that generator)."""
    import contextlib
    import io
    import machine
    import movie
    import palette
    import screen
    import video
    import video_mode

    out = {}
    for tag, mode_name, every_n, n_frames, n_audio, dseed, rseed, pal_id in (
            ("DHGR_n1", "DHGR", 1, 30, 30 * 490 + 200, 31, 4, 5),
            ("DHGR_n2", "DHGR", 2, 30, 30 * 490 + 200, 31, 5, 5),
            ("HGR_n1", "HGR", 1, 30, 30 * 490 + 200, 32, 6, 5),
            ("HGR_n2", "HGR", 2, 30, 30 * 490 + 200, 32, 7, 5),
            # audio runs out first (mid-frame, on a frame that is not encoded)
            ("DHGR_n2_audio_end", "DHGR", 2, 8, 5 * 490 + 77, 33, 8, 0),
    ):
        t = time.time()
        mode = video_mode.VideoMode[mode_name]
        pal = palette.Palette(pal_id)
        frames = synth_frames(mode_name, n_frames, dseed, coherent=True)

        class Audio:
            sample_rate = 14700.

            def audio_stream(self):
                for _ in range(n_audio):
                    yield 0

        class Grabber:
            input_frame_rate = 30
            video_mode = mode
            served = []

            def frames(self):
                for f in range(n_frames):
                    self.served.append(f)
                    main = screen.MemoryMap(screen_page=1, page_offset=frames[f, 0].copy())
                    main.frame_index = f   # (a tag for the recorder below; the reference never looks at it)
                    aux = None
                    if mode_name == "DHGR":
                        aux = screen.MemoryMap(screen_page=1, page_offset=frames[f, 1].copy())
                    yield main, aux

        Grabber.served = []
        random.seed(rseed)
        np.random.seed(rseed)
        m = movie.Movie.__new__(movie.Movie)
        m.filename = None
        m.every_n_video_frames = every_n
        m.max_bytes_out = None
        m.video_mode = mode
        m.palette = pal
        m.audio = Audio()
        m.frame_grabber = Grabber()
        m.video = video.Video(m.frame_grabber, ticks_per_second=m.audio.sample_rate, mode=mode, palette=pal)
        m.stream_pos = 0
        m.ticks = 0
        m.state = machine.Machine()
        m.aux_memory_bank = False

        calls = []   # [frame index, is_aux, pulled]
        real_encode_frame = m.video.encode_frame

        def recording_encode_frame(target, is_aux, _calls=calls, _real=real_encode_frame):
            rec = [target.main_memory.frame_index, int(bool(is_aux)), 0]
            _calls.append(rec)
            gen = _real(target, is_aux=is_aux)

            def counted():
                for item in gen:
                    rec[2] += 1
                    yield item
            return counted()

        m.video.encode_frame = recording_encode_frame
        with contextlib.redirect_stdout(io.StringIO()):
            stream = np.array(list(m.emit_stream(m.encode())), dtype=np.uint8)
        v = m.video
        out[tag + "/frames"] = frames
        out[tag + "/meta"] = np.array([0 if mode_name == "HGR" else 1, pal_id, rseed, every_n, n_audio, m.ticks,
                                       v.frame_number], dtype=np.int32)
        out[tag + "/calls"] = np.array(calls, dtype=np.int32).reshape(-1, 3)
        out[tag + "/stream"] = stream
        out[tag + "/mem_main"] = v.memory_map.page_offset.copy()
        out[tag + "/up_main"] = v.update_priority.copy()
        if mode_name == "DHGR":
            out[tag + "/mem_aux"] = v.aux_memory_map.page_offset.copy()
            out[tag + "/up_aux"] = v.aux_update_priority.copy()
        out[tag + "/py_next"] = np.array([random.getrandbits(8) for _ in range(4)], dtype=np.uint8)
        out[tag + "/np_next"] = np.random.randint(0, 256, size=4).astype(np.uint8)
        print("%s: %d calls, %d bytes, ticks %d, frame_number %d (%.1fs)" % (
            tag, len(calls), len(stream), m.ticks, v.frame_number, time.time() - t))
    np.savez_compressed(os.path.join(HERE, "g7_movie.npz"), **out)


if __name__ == "__main__" and "--movie-only" in sys.argv:
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    setup_reference("/tmp/iiv_ref")
    make_movie_golden("/tmp/iiv_ref")


def make_fourth_offset_golden(scratch):
    """g8: the reference's own encode loop with ONE literal changed -- the exit test of the extra-offset loop,
    `if len(offsets) == 3: break` (video.py:180-181), reads 4 -- i.e. what video.py:146 ("Need to find 3 more
    offsets to fill this opcode") announces.  The changed method is built here, at generation time, from the
    imported reference's source (inspect.getsource + one str.replace) and bound to the imported class; only inputs
    and outputs go into the fixture.  This is what pins the oracle's orc_video_set_fourth_offset()."""
    import inspect
    import textwrap
    import video
    src = textwrap.dedent(inspect.getsource(video.Video._index_changes))
    old = "if len(offsets) == 3:"
    assert src.count(old) == 1, "the reference's exit test moved"
    ns = dict(vars(video))
    exec(compile(src.replace(old, "if len(offsets) == 4:"), "<_index_changes, exit test at 4>", "exec"), ns)
    original = video.Video._index_changes
    video.Video._index_changes = ns["_index_changes"]
    try:
        g8 = {}
        cases = [("DHGR_iid_s1", "DHGR", 5, 3, 7, False, 1, None), ("HGR_iid_s2", "HGR", 5, 3, 7, False, 2, None),
                 ("DHGR_coh_s3", "DHGR", 5, 5, 7, True, 3, None), ("HGR_coh_s1", "HGR", 0, 4, 11, True, 1, None),
                 ("HGR_exhaust", "HGR", 5, 2, 21, False, 5, [(0, 0, 5200), (1, 0, 300), (1, 0, 5600)]),
                 ("DHGR_exhaust", "DHGR", 5, 2, 22, False, 5,
                  [(0, 0, 4200), (0, 1, 4300), (0, 0, 900), (1, 1, 2500), (1, 0, 2500), (1, 1, 3000)])]
        for (tag, mode_name, pal, nf, dseed, coh, rseed, sched) in cases:
            t = time.time()
            frames = synth_frames(mode_name, nf, dseed, coherent=coh)
            if sched is None:
                sched = movie_schedule(mode_name, nf)
            st = run_reference(mode_name, pal, frames, sched, rseed, rseed)
            g8[tag + "/frames"] = frames
            g8[tag + "/schedule"] = np.array(sched, dtype=np.int32)
            g8[tag + "/meta"] = np.array([0 if mode_name == "HGR" else 1, pal, rseed, rseed], dtype=np.int32)
            for k, val in st.items():
                g8[tag + "/" + k] = val
            distinct = np.mean([len(set(r[2:6])) for r in st["ops"].tolist()])
            print("%s: %d ops, %.2f distinct offsets per opcode, sha %s (%.1fs)" % (
                tag, len(st["ops"]), distinct, sha(st["ops"])[:16], time.time() - t))
        np.savez_compressed(os.path.join(HERE, "g8_fourth_offset.npz"), **g8)
        print("g8 written")
    finally:
        video.Video._index_changes = original


if __name__ == "__main__" and "--fourth-only" in sys.argv:
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    setup_reference("/tmp/iiv_ref")
    make_fourth_offset_golden("/tmp/iiv_ref")
