"""Byte emission (f2): oracle vs the reference's own Movie.emit_stream output (CPU),
stream-length arithmetic of the C ABI (CPU, no device work), and the HIP kernel vs
golden and oracle (GPU)."""

import io

import numpy as np
import pytest


def _tags(g):
    return sorted(set(k.split("/")[0] for k in g.files if "/" in k))


def test_oracle_emit_stream_matches_reference(O, golden):
    g = golden.g6_a2m
    for t in _tags(g):
        mode, mx = (int(x) for x in g[t + "/meta"])
        got = O.emit_stream(mode, g[t + "/ops"], g[t + "/ticks"], g["tick_addr"], g["special_addr"][0],
                            g["special_addr"][1], None if mx < 0 else mx)
        assert np.array_equal(got, g[t + "/stream"]), t


def test_stream_length_without_gpu(native, O, golden):
    import a2m
    g = golden.g6_a2m
    addr = a2m.OpcodeAddresses(g["tick_addr"], g["special_addr"][0], g["special_addr"][1], g["special_addr"][2])
    for t in _tags(g):
        mode, mx = (int(x) for x in g[t + "/meta"])
        assert a2m.stream_length(mode, len(g[t + "/ops"]), addr, None if mx < 0 else mx) == len(g[t + "/stream"])
    for n in (0, 1, 290, 291, 292, 583, 5000):
        for mx in (None, 1, 7, 8, 2044, 2048, 2049, 10000):
            want = len(O.emit_stream(1, np.zeros((n, 6), np.uint8) + 32, np.full(n, 4, np.uint8), g["tick_addr"], 1, 2, mx))
            assert a2m.stream_length(1, n, addr, mx) == want, (n, mx)


def test_symbol_table_parser():
    import a2m
    import symbol_table
    lines = ['version\tmajor=2,minor=0', 'sym\tid=3,name="op_ack",addrsize=absolute,val=0xBA72,type=lab',
             'sym\tid=6,name="op_terminate",val=0xBA64,type=lab', 'sym\tid=9,name="op_nop",val=0x4070,type=lab',
             'sym\tid=7,name="other",val=0x1234,type=lab']
    for ti, t in enumerate(range(4, 68, 2)):
        for page in range(32, 64):
            lines.append('sym\tid=1,name="op_tick_%d_page_%d",val=0x%X,type=lab' % (t, page, 0x8000 + ti * 32 + page))
    syms = symbol_table.SymbolTable().parse(io.StringIO("\n".join(lines)))
    assert syms['"op_ack"']["val"] == "0xBA72" and '"other"' in syms
    import tempfile, os
    with tempfile.NamedTemporaryFile("w", suffix=".dbg", delete=False) as f:
        f.write("\n".join(lines))
    try:
        a = a2m.OpcodeAddresses.from_debug_file(f.name)
    finally:
        os.unlink(f.name)
    assert a.ack == 0xBA72 and a.terminate == 0xBA64 and a.tick[3, 5] == 0x8000 + 3 * 32 + 37
    with pytest.raises(ValueError):
        with tempfile.NamedTemporaryFile("w", suffix=".dbg", delete=False) as f:
            f.write("\n".join(lines[:5]))
        a2m.OpcodeAddresses.from_debug_file(f.name)


@pytest.mark.gpu
def test_emit_kernel_matches_reference_and_oracle(native, O, golden):
    import torch
    import a2m
    g = golden.g6_a2m
    addr = a2m.OpcodeAddresses(g["tick_addr"], g["special_addr"][0], g["special_addr"][1], g["special_addr"][2])
    for t in _tags(g):
        mode, mx = (int(x) for x in g[t + "/meta"])
        ops = torch.from_numpy(g[t + "/ops"][None].copy()).cuda()
        ticks = torch.from_numpy(g[t + "/ticks"][None].copy()).cuda()
        got = a2m.emit_stream(mode, ops, ticks, addr, None if mx < 0 else mx).cpu().numpy()[0]
        assert np.array_equal(got, g[t + "/stream"]), t
    # a batch of independent streams, several 2 KiB frames each
    rng = np.random.default_rng(0)
    S, n = 37, 2500
    ops = rng.integers(0, 256, (S, n, 6)).astype(np.uint8)
    ops[:, :, 0] = rng.integers(32, 64, (S, n))
    ticks = (rng.integers(0, 32, (S, n)) * 2 + 4).astype(np.uint8)
    for mode in (0, 1):
        for mx in (None, 9000):
            got = a2m.emit_stream(mode, torch.from_numpy(ops).cuda(), torch.from_numpy(ticks).cuda(), addr, mx).cpu().numpy()
            for s in (0, 17, 36):
                want = O.emit_stream(mode, ops[s], ticks[s], addr.tick, addr.ack, addr.terminate, mx)
                assert np.array_equal(got[s], want), (mode, mx, s)


@pytest.mark.gpu
def test_encode_then_emit_end_to_end(native, O, oracle_tables, device_tables, golden):
    """encode (P3) -> emit (f2) on the device equals oracle encode -> oracle emit."""
    import torch
    import a2m
    import stream_batch
    g = golden.g6_a2m
    addr = a2m.OpcodeAddresses(g["tick_addr"], g["special_addr"][0], g["special_addr"][1], g["special_addr"][2])
    t, s = device_tables.get(1)
    fm, fa = stream_batch.synth_frames_torch(3, 2, True, seed=9)
    b = stream_batch.StreamBatch(1, t, s, 3, seeds=[(1, 1), (2, 2), (3, 3)], dm=device_tables.dm[(1, 5)])
    ops, segs = b.encode_frames(fm, fa, 2)
    b.enc.check()
    n = ops.shape[1]
    ticks = torch.full((3, n), 34, dtype=torch.uint8, device="cuda")   # silence: au = 0 -> tick 34 (movie.py:104-107)
    stream = a2m.emit_stream(1, ops, ticks, addr).cpu().numpy()
    for i in range(3):
        v = O.Video(1, oracle_tables.get(1), seed_py=i + 1, seed_np=i + 1)
        exp = []
        for (f, ia, _, k) in segs:
            v.encode_frame(fm[i, f].cpu().numpy(), fa[i, f].cpu().numpy(), ia)
            exp.append(v.next(k))
        want = O.emit_stream(1, np.concatenate(exp), np.full(n, 34, np.uint8), addr.tick, addr.ack, addr.terminate)
        assert np.array_equal(stream[i], want)
    b.close()


@pytest.mark.gpu
def test_emit_kernel_random_sizes_and_cuts(native, O, golden):
    """Random opcode counts around the 2 KiB socket-frame boundaries, random ticks, random max_bytes_out cuts
    (movie.py:132-134), several streams at once, both modes: the kernel's bytes equal the oracle's restatement of
    Movie.emit_stream (which test_oracle_emit_stream_matches_reference pins to the reference's own output)."""
    import torch
    import a2m
    g = golden.g6_a2m
    addr = a2m.OpcodeAddresses(g["tick_addr"], g["special_addr"][0], g["special_addr"][1], g["special_addr"][2])
    rng = np.random.default_rng(77)
    for trial in range(60):
        mode = int(rng.integers(0, 2))
        n = int(rng.choice([0, 1, 2, 290, 291, 292, 293, 582, 583, 584, int(rng.integers(1, 4000))]))
        S = int(rng.integers(1, 5))
        ops = rng.integers(0, 256, (S, n, 6), dtype=np.uint8)
        ops[:, :, 0] = rng.integers(32, 64, (S, n))
        ticks = (rng.integers(2, 34, (S, n)) * 2).astype(np.uint8)
        mx = None if rng.random() < 0.5 else int(rng.choice([1, 7, 8, 14, 2043, 2044, 2048, 2049, int(rng.integers(1, 7 * n + 64))]))
        got = a2m.emit_stream(mode, torch.from_numpy(ops).cuda(), torch.from_numpy(ticks).cuda(), addr, mx).cpu().numpy()
        for s in range(S):
            want = O.emit_stream(mode, ops[s], ticks[s], g["tick_addr"], addr.ack, addr.terminate, mx)
            assert got.shape[1] == len(want) and (got[s] == want).all(), (trial, mode, n, S, mx, s)
