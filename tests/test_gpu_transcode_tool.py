"""GPU: tools/transcode_clip.py -- RGB frames -> memory maps -> Movie-paced encode -> .a2m bytes, all on the device --
against the same chain through the oracle (ingest definition, encode restatement, emit restatement): the file's bytes
are equal.  Rows f3 -> hot path + f1 -> f2 of SURVEY 8 in one run, in both modes, with and without the fourth offset."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mode_name,fourth", [("DHGR", False), ("HGR", False), ("DHGR", True)])
def test_transcode_clip_equals_the_oracle_chain(tmp_path, O, oracle_tables, mode_name, fourth):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import stream_batch
    import transcode_clip
    n = 5
    out = tmp_path / "clip.a2m"
    npy = tmp_path / "clip.npy"
    rgb = transcode_clip.test_card(n)
    np.save(npy, rgb)
    args = [sys.executable, os.path.join(ROOT, "tools", "transcode_clip.py"), "--frames", str(npy), "--out", str(out),
            "--mode", mode_name, "--seed", "7", "--tick", "20"] + (["--fourth"] if fourth else [])
    r = subprocess.run(args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "NOT playable" in r.stderr          # (no --dbg: placeholder addresses, said so)
    got = np.frombuffer(out.read_bytes(), np.uint8)

    mode = 1 if mode_name == "DHGR" else 0
    maps = [O.frame_to_memory_map(mode, O.PALETTE_RGB[5], rgb[f], O.DITHER_DIFFUSION) for f in range(n)]
    v = O.Video(mode, oracle_tables.get(mode, 5), seed_py=7, seed_np=7)
    v.set_fourth_offset(fourth)
    ops = []
    for (fr, ia, restart, k) in stream_batch.MovieClock(mode == 1).segments(n):
        if restart:
            v.encode_frame(maps[fr][0], maps[fr][1] if mode == 1 else None, ia)
        ops.append(v.next(k))
    ops = np.concatenate(ops)
    tick_addr = (0x8000 + 16 * np.arange(1024)).astype(np.uint16)
    exp = O.emit_stream(mode, ops, np.full(len(ops), 20, np.uint8), tick_addr, 0xc000, 0xc100)
    assert len(got) == len(exp) and len(got) % 2048 == 0
    assert (got == exp).all()
