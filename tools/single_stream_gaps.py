"""Where one clip's time goes: kernel time (rocprofv3-free: HIP events around every launch) against wall time of a
50-frame driver step -- the difference is kernel boundaries + host.  python tools/single_stream_gaps.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ii-vision_amd", "transcoder"))
import torch
import _iiv_native as native, stream_batch, palette
for mode in (native.DHGR, native.HGR):
    _, dm = native.cie2000_matrix(palette.NTSCPalette.rgb_array())
    table = native.build_table(mode, dm, True); store = native.build_store_table(mode, dm)
    fm, fa = stream_batch.synth_frames_torch(1, 120, mode == native.DHGR, seed=99)
    for prof in (False, True):
        b = stream_batch.StreamBatch(mode, table, store, 1, seeds=[(1, 1)], dm=dm)
        b.encode_frames(fm, fa, 10)
        if prof:
            b.enc.profile(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, segs = b.encode_frames(fm, fa, 50)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        line = "%s: 50 frames in %.2f ms (%.0f frames/s); host returned after %.2f ms; %d launch rounds" % (
            "DHGR" if mode == native.DHGR else "HGR", 1e3 * dt, 50 / dt, 1e3 * (t1 - t0), len(segs))
        if prof:
            p = b.enc.profile_read()
            k = p["prologue_ms"] + p["greedy_ms"]
            line += "; with events: prologue %.2f ms in %d launches (%.1f us each), team kernel %.2f ms in %d (%.1f us each); kernels %.2f ms = %.0f %% of the wall time" % (
                p["prologue_ms"], p["prologue_launches"], 1e3 * p["prologue_ms"] / max(p["prologue_launches"], 1),
                p["greedy_ms"], p["greedy_launches"], 1e3 * p["greedy_ms"] / max(p["greedy_launches"], 1), k, 100 * k / (1e3 * dt))
        print(line, flush=True)
        b.close()
