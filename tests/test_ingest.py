"""f3: RGB frame -> memory map.  The reference delegates this to the external
bmp2dhr tool, so there is no reference output: include/iivision.h specifies the conversion and the oracle restates it
(orc_frame_to_memory_map).  CPU: the specification is consistent with the colour model that IS pinned
to the reference (solid colours survive the round trip through pack -> mask -> colour model).
GPU: the HIP kernel equals the definition bit for bit; the host mirror yields memory maps."""

import numpy as np
import pytest


def _test_frames(n, seed):
    """gradients, colour bars and noise: (n, 192, 280, 3) u8"""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:192, 0:280]
    out = np.zeros((n, 192, 280, 3), np.uint8)
    for i in range(n):
        kind = i % 4
        if kind == 0:
            out[i] = rng.integers(0, 256, (192, 280, 3))
        elif kind == 1:
            out[i, ..., 0] = (x * 255 // 279)
            out[i, ..., 1] = (y * 255 // 191)
            out[i, ..., 2] = ((x + y + 7 * i) % 256)
        elif kind == 2:
            bars = rng.integers(0, 256, (8, 3))
            out[i] = bars[(x * 8 // 280)]
        else:
            out[i] = (rng.integers(0, 256, (24, 35, 3)).repeat(8, axis=0).repeat(8, axis=1))
    return out


@pytest.mark.parametrize("mode", [1, 0])
def test_solid_colours_round_trip_through_the_colour_model(O, mode):
    """A frame filled with a palette colour becomes dots whose windows the (reference-pinned) colour
    model reads back as that colour: the dot patterns and phases of the conversion are right."""
    pal = O.PALETTE_RGB[0]     # //gs: sixteen distinct colours
    holes = O.screen_holes()
    L = O.lib()
    for c in (range(16) if mode == 1 else (0, 3, 12, 15, 6, 9)):
        rgb = np.tile(pal[c], (192, 280, 1)).astype(np.uint8)
        main, aux = O.frame_to_memory_map(mode, pal, rgb, 0)
        assert (main[holes] == 0).all() and (aux is None or (aux[holes] == 0).all())
        packed = O.pack(mode, main, aux)
        for page in (0, 9, 31):
            for col in (1, 7, 18):          # interior columns of the first 40-byte segment
                for o in range(O.num_offsets(mode)):
                    w = L.orc_mask_and_shift(mode, int(packed[page, col]), o)
                    assert (O.pixel_values(mode, w, o) == c).all(), (mode, c, page, col, o)


def test_ingest_definition_properties(O):
    pal = O.PALETTE_RGB[5]
    rgb = _test_frames(4, 1)
    for mode in (0, 1):
        for f in range(4):
            main, aux = O.frame_to_memory_map(mode, pal, rgb[f], 0)
            if mode == 1:
                assert (main < 128).all() and (aux < 128).all()      # DHGR bytes carry no palette bit (video.py:137)
            again, _ = O.frame_to_memory_map(mode, pal, rgb[f], 0)
            assert (again == main).all()
    # the dither changes flat mid-tone areas, not areas that sit on a palette colour
    flat = np.full((192, 280, 3), 128, np.uint8)
    a, _ = O.frame_to_memory_map(1, pal, flat, 0)
    b, _ = O.frame_to_memory_map(1, pal, flat, 64)
    assert (a != b).any()
    white = np.full((192, 280, 3), 255, np.uint8)
    a, _ = O.frame_to_memory_map(1, pal, white, 0)
    b, _ = O.frame_to_memory_map(1, pal, white, 16)
    assert (a == b).all()


def _chosen_colours(O, mode, pal, main, aux):
    """The palette colours a conversion chose, decoded back from the memory maps: (192, 140, 3) RGB.  DHGR: the
    aligned dot quad of a colour pixel IS its colour value (colours.py:100-134); HGR: the 2-dot pattern of a pixel
    under the palette bit of the byte holding its first dot (black, violet | blue, green | orange, white)."""
    out = np.zeros((192, 140, 3), np.int64)
    hgr = np.array([[0, 3, 12, 15], [0, 6, 9, 15]])
    for y in range(192):
        base = O.lib().orc_y_to_base_addr(y, 0) - 0x2000
        if mode == 1:
            row = np.zeros(80, np.int64)
            row[0::2] = aux.reshape(-1)[base:base + 40]
            row[1::2] = main.reshape(-1)[base:base + 40]
            dots = ((row[:, None] >> np.arange(7)) & 1).reshape(-1)                 # 560 dots
            val = (dots.reshape(140, 4) << np.arange(4)).sum(axis=1)
        else:
            row = main.reshape(-1)[base:base + 40].astype(np.int64)
            dots = ((row[:, None] >> np.arange(7)) & 1).reshape(-1)                 # 280 dots
            patt = dots[0::2] | (dots[1::2] << 1)
            pb = (row >> 7)[(2 * np.arange(140)) // 7]
            val = hgr[pb, patt]
        out[y] = pal[val]
    return out


def _block_error(O, mode, pal, rgb, dither):
    """Quality metric of a conversion: a dither trades pixel-exact colour for the right colour ON AVERAGE, so the
    error is measured after a 4 x 4 block mean over the 140 x 192 colour pixels -- root mean square over the blocks
    of the weighted RGB distance (the conversion's own 2 dr^2 + 4 dg^2 + 3 db^2) between the source's block mean
    and the block mean of the colours chosen."""
    main, aux = O.frame_to_memory_map(mode, pal, rgb, dither)
    got = _chosen_colours(O, mode, np.asarray(pal, np.int64), main, aux).astype(np.float64)
    src = (rgb[:, 0::2].astype(np.float64) + rgb[:, 1::2]) / 2
    def blocks(a):
        return a.reshape(48, 4, 35, 4, 3).mean(axis=(1, 3))
    d = blocks(got) - blocks(src)
    return float(np.sqrt((d ** 2 * np.array([2.0, 4.0, 3.0])).sum(axis=-1).mean()))


def test_error_diffusion_beats_ordered_dither_on_average_colour(O):
    """the reference asks its external tool for an error-diffusion dither (bmp2dhr D9,
    frame_grabber.py:80-82); IIV_DITHER_DIFFUSION is this build's (Floyd-Steinberg, integer; bmp2dhr itself cannot be
    matched here).  On smooth content -- a two-axis colour gradient, a grey ramp, a soft vignette -- its block-mean
    colour error is below the ordered dither's and far below no dither's, in both modes; on content that already sits
    on palette colours it changes nothing."""
    pal = O.PALETTE_RGB[5]
    y, x = np.mgrid[0:192, 0:280]
    grad = np.stack([x * 255 // 279, y * 255 // 191, (x + y) * 255 // 470], -1).astype(np.uint8)
    ramp = np.repeat((x * 255 // 279)[..., None], 3, -1).astype(np.uint8)
    r2 = ((x - 140.0) / 140) ** 2 + ((y - 96.0) / 96) ** 2
    vign = np.clip(np.stack([230 - 120 * r2, 140 - 60 * r2, 90 + 100 * r2], -1), 0, 255).astype(np.uint8)
    report = []
    for mode in (1, 0):
        for name, img in (("gradient", grad), ("grey ramp", ramp), ("vignette", vign)):
            e_none = _block_error(O, mode, pal, img, 0)
            e_ord = _block_error(O, mode, pal, img, 32)
            e_dif = _block_error(O, mode, pal, img, O.DITHER_DIFFUSION)
            report.append("%s %-9s none %6.1f  ordered(32) %6.1f  diffusion %6.1f" % ("DHGR" if mode else "HGR ", name, e_none, e_ord, e_dif))
            assert e_dif < e_ord < e_none, report[-1]
            assert e_dif < 0.6 * e_none, report[-1]
    print("\nblock-mean colour error (rms, weighted RGB):\n" + "\n".join(report))
    solid = np.tile(np.asarray(pal[12], np.uint8), (192, 280, 1))
    for mode in (1, 0):
        a = O.frame_to_memory_map(mode, pal, solid, 0)
        b = O.frame_to_memory_map(mode, pal, solid, O.DITHER_DIFFUSION)
        assert (a[0] == b[0]).all() and (mode == 0 or (a[1] == b[1]).all())


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 0])
@pytest.mark.parametrize("pal_id", [5, 0])
def test_ingest_kernel_equals_definition(native, O, mode, pal_id):
    import torch
    rgb = _test_frames(9, 2 + mode)
    for dither in (0, 32, 255, native.DITHER_DIFFUSION):
        main, aux = native.frames_to_memory_maps(mode, O.PALETTE_RGB[pal_id], torch.from_numpy(rgb).cuda(), dither)
        main = main.cpu().numpy()
        aux = aux.cpu().numpy() if aux is not None else None
        for f in range(len(rgb)):
            em, ea = O.frame_to_memory_map(mode, O.PALETTE_RGB[pal_id], rgb[f], dither)
            assert (main[f] == em).all(), (mode, pal_id, dither, f)
            if mode == 1:
                assert (aux[f] == ea).all(), (mode, pal_id, dither, f)


@pytest.mark.gpu
def test_frame_grabber_feeds_the_encoder(native, O, oracle_tables, device_tables):
    """frame_grabber.ArrayFrameGrabber -> memory maps -> StreamBatch: an RGB clip transcodes end to
    end on the device, and the opcodes equal the oracle's on the oracle-converted frames."""
    import frame_grabber
    import palette
    import stream_batch
    import video_mode
    rgb = _test_frames(6, 9)
    fg = frame_grabber.ArrayFrameGrabber(rgb, video_mode.VideoMode.DHGR, palette.Palette.NTSC, dither=32)
    assert fg.input_frame_rate == 30 and fg.video_mode == video_mode.VideoMode.DHGR
    maps = list(fg.frames())
    assert len(maps) == 6 and maps[0][0].page_offset.shape == (32, 256) and maps[0][1] is not None
    main, aux = fg.memory_maps()
    for f in range(6):
        em, ea = O.frame_to_memory_map(1, O.PALETTE_RGB[5], rgb[f], 32)
        assert (maps[f][0].page_offset == em).all() and (maps[f][1].page_offset == ea).all()
    t, s = device_tables.get(1)
    b = stream_batch.StreamBatch(1, t, s, 1, seeds=[(3, 3)], dm=device_tables.dm[(1, 5)])
    ops, segs = b.encode_frames(main[None].contiguous(), aux[None].contiguous(), 6)
    b.enc.check()
    v = O.Video(1, oracle_tables.get(1), seed_py=3, seed_np=3)
    exp = []
    for (f, ia, restart, k) in segs:
        if restart:
            v.encode_frame(maps[f][0].page_offset, maps[f][1].page_offset, ia)
        exp.append(v.next(k))
    assert (ops.cpu().numpy()[0] == np.concatenate(exp)).all()
    b.close()
    with pytest.raises(ValueError):
        frame_grabber.ArrayFrameGrabber(np.zeros((2, 192, 140, 3), np.uint8), video_mode.VideoMode.HGR)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 0])
def test_ingest_kernel_on_noise_and_extremes(native, O, mode):
    """Uniform noise, saturated primaries, black, white, a one-pixel checkerboard and steep ramps -- the inputs that push
    the error-diffusion accumulators and the dither offsets to their clamps -- through every dither kind and both
    palettes: kernel = definition, byte for byte."""
    import torch
    rng = np.random.default_rng(12)
    y, x = np.mgrid[0:192, 0:280]
    frames = [rng.integers(0, 256, (192, 280, 3), dtype=np.uint8), np.zeros((192, 280, 3), np.uint8),
              np.full((192, 280, 3), 255, np.uint8)]
    for c in ((255, 0, 0), (0, 255, 0), (0, 0, 255), (255, 255, 0)):
        frames.append(np.broadcast_to(np.array(c, np.uint8), (192, 280, 3)).copy())
    frames.append((((x + y) & 1) * 255).astype(np.uint8)[..., None].repeat(3, axis=2))
    frames.append(np.stack([(x * 255 // 279), (y * 255 // 191), ((x * 7 + y * 13) % 256)], axis=2).astype(np.uint8))
    frames.append(np.where(rng.random((192, 280, 1)) < 0.5, 0, 255).astype(np.uint8).repeat(3, axis=2))
    rgb = np.ascontiguousarray(np.stack(frames))
    for pal_id in (5, 0):
        for dither in (0, 1, 17, 128, 255, native.DITHER_DIFFUSION):
            main, aux = native.frames_to_memory_maps(mode, O.PALETTE_RGB[pal_id], torch.from_numpy(rgb).cuda(), dither)
            main = main.cpu().numpy()
            aux = aux.cpu().numpy() if aux is not None else None
            for f in range(len(rgb)):
                em, ea = O.frame_to_memory_map(mode, O.PALETTE_RGB[pal_id], rgb[f], dither)
                assert (main[f] == em).all(), (mode, pal_id, dither, f)
                if mode == 1:
                    assert (aux[f] == ea).all(), (mode, pal_id, dither, f)


@pytest.mark.gpu
def test_ingest_writes_into_a_batch_slice_asynchronously(native, O):
    """out=: the conversion writes straight into a (streams, frames, 32, 256) slice of a batch's target frames, several
    calls back to back on one stream without a host synchronisation in between (the pipeline of bench.py's e2e leg);
    every frame, holes included, equals the definition -- the buffers start as 0xff, so an unwritten byte shows."""
    import torch
    rgb = _test_frames(12, 21)
    dev = torch.from_numpy(rgb).cuda()
    for mode in (1, 0):
        main = torch.full((6, 4, 32, 256), 255, dtype=torch.uint8, device="cuda")
        aux = torch.full((6, 4, 32, 256), 255, dtype=torch.uint8, device="cuda") if mode == 1 else None
        for dither, s0 in ((32, 0), (native.DITHER_DIFFUSION, 3)):      # streams 0..2 ordered, 3..5 diffusion
            native.frames_to_memory_maps(mode, O.PALETTE_RGB[5], dev, dither,
                                         out=(main[s0:s0 + 3], aux[s0:s0 + 3] if aux is not None else None))
        torch.cuda.synchronize()
        m, a = main.cpu().numpy().reshape(24, 32, 256), (aux.cpu().numpy().reshape(24, 32, 256) if aux is not None else None)
        for i in range(24):
            dither = 32 if i < 12 else O.DITHER_DIFFUSION
            em, ea = O.frame_to_memory_map(mode, O.PALETTE_RGB[5], rgb[i % 12], dither)
            assert (m[i] == em).all(), (mode, i)
            if mode == 1:
                assert (a[i] == ea).all(), (mode, i)
    with pytest.raises(ValueError):
        native.frames_to_memory_maps(1, O.PALETTE_RGB[5], dev, 0, out=(main[:2], None))
