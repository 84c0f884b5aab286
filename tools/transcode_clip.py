"""End to end on one GPU, everything on the device: RGB frames -> memory maps (f3, csrc/iiv_ingest.hip) ->
Movie-paced encode (the hot path + f1: prologue / greedy kernels driven by stream_batch.MovieClock) -> player byte
stream (f2, csrc/iiv_a2m.hip) -> an .a2m file.  What the reference's `main.py in.mp4 out.a2m` does between its decoder
and its output file (transcoder/main.py, movie.py:56-161), minus audio: every opcode carries the same speaker duty
cycle (`--tick`, 4..66 even: movie.py:104-107).

    python tools/transcode_clip.py --frames clip.npy --out clip.a2m --dbg /path/to/player/iivision.dbg
    python tools/transcode_clip.py --synthetic 90 --out /tmp/bars.a2m            # a moving test card

--frames: uint8 array (n, 192, 280, 3), i.e. what the reference's FileFrameGrabber holds after its resize
(frame_grabber.py:75).  --dbg: the player's cc65 debug file, from which the opcode entry points are read exactly as
opcodes._parse_symbol_table does (opcodes.py:168-185); without it the stream is written with placeholder addresses
and is NOT playable (the tool says so).  --fourth / --joint: the two optional quality modes."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ii-vision_amd", "transcoder"))
import numpy as np  # noqa: E402


def test_card(n):
    """n frames (192, 280, 3): colour bars drifting over a grey ramp"""
    y, x = np.mgrid[0:192, 0:280]
    bars = np.array([[255, 255, 255], [255, 255, 0], [0, 255, 255], [0, 255, 0], [255, 0, 255], [255, 0, 0], [0, 0, 255], [0, 0, 0]], np.uint8)
    out = np.empty((n, 192, 280, 3), np.uint8)
    for f in range(n):
        out[f] = bars[((x + 3 * f) // 35) % 8]
        out[f, 128:] = ((x[128:] + y[128:] - 2 * f) % 256)[..., None]
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--frames", help=".npy file, uint8 (n, 192, 280, 3)")
    ap.add_argument("--synthetic", type=int, default=0, help="instead of --frames: this many frames of a moving test card")
    ap.add_argument("--out", required=True)
    ap.add_argument("--mode", choices=["DHGR", "HGR"], default="DHGR")
    ap.add_argument("--palette", choices=["NTSC", "IIGS"], default="NTSC")
    ap.add_argument("--dither", default="diffusion", help='"diffusion" (Floyd-Steinberg) or the amplitude 0..255 of the ordered dither')
    ap.add_argument("--dbg", help="player/iivision.dbg (opcode entry points)")
    ap.add_argument("--tick", type=int, default=34, help="speaker duty cycle of every opcode (4..66, even)")
    ap.add_argument("--fourth", action="store_true", help="IIV_OPT_FOURTH_OFFSET (not the reference's stream)")
    ap.add_argument("--joint", action="store_true", help="IIV_CONTENT_JOINT (not the reference's stream)")
    ap.add_argument("--seed", type=int, default=1, help="random.seed / np.random.seed of the encoder's two nonce streams")
    a = ap.parse_args()
    if a.tick < 4 or a.tick > 66 or a.tick % 2:
        ap.error("--tick: 4..66, even")
    if bool(a.frames) == bool(a.synthetic):
        ap.error("one of --frames / --synthetic")

    import torch
    import _iiv_native as native
    import a2m
    import frame_grabber
    import palette
    import stream_batch
    import video_mode

    rgb = np.load(a.frames) if a.frames else test_card(a.synthetic)
    mode = native.DHGR if a.mode == "DHGR" else native.HGR
    pal_id = palette.Palette.NTSC if a.palette == "NTSC" else palette.Palette.IIGS
    t0 = time.perf_counter()
    grab = frame_grabber.ArrayFrameGrabber(rgb, video_mode.VideoMode[a.mode], pal_id,
                                           dither=a.dither if a.dither == "diffusion" else int(a.dither))
    main_maps, aux_maps = grab.memory_maps()                       # (n, 32, 256) on the device
    _, dm = native.cie2000_matrix(palette.PALETTES[pal_id].rgb_array())
    table = native.build_table(mode, dm, True)
    store = native.build_store_table(mode, dm)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    batch = stream_batch.StreamBatch(mode, table, store, 1, seeds=[(a.seed, a.seed)], dm=dm, joint_content=a.joint, fourth_offset=a.fourth,
                                     input_frame_rate=grab.input_frame_rate)
    n = int(main_maps.shape[0])
    ops, segs = batch.encode_frames(main_maps[None], aux_maps[None] if aux_maps is not None else None, n)
    batch.enc.check()
    if a.dbg:
        addr = a2m.OpcodeAddresses.from_debug_file(a.dbg)
    else:
        print("no --dbg: placeholder opcode addresses -- the stream has the right layout but is NOT playable", file=sys.stderr)
        addr = a2m.OpcodeAddresses(0x8000 + 16 * np.arange(1024, dtype=np.uint16).reshape(32, 32), 0xc000, 0xc100)
    ticks = torch.full((1, ops.shape[1]), a.tick, dtype=torch.uint8, device="cuda")
    stream = a2m.emit_stream(mode, ops, ticks, addr)
    data = stream[0].cpu().numpy().tobytes()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    with open(a.out, "wb") as f:
        f.write(data)
    n_ops = int(ops.shape[1])
    distinct = np.mean([len(set(r[2:6])) for r in ops[0].cpu().numpy().tolist()])
    print("%d frames (%s, %s palette, dither %s) -> %d opcodes (%.2f distinct offsets each), %d generators -> %d bytes in %s" % (
        n, a.mode, a.palette, a.dither, n_ops, distinct, len(stream_batch.merge_generators(segs)), len(data), a.out))
    print("tables + ingest %.2f s, encode + emit %.3f s = %.0f frames/s for this one clip (many clips at once: bench.py)" % (
        t1 - t0, t2 - t1, n / (t2 - t1)))
    batch.close()


if __name__ == "__main__":
    main()
