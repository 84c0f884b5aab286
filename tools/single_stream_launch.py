"""One clip alone: frames/s of the device work (the Python MovieClock is run outside the timed region),
for the one-wave kernel and the team kernel.  (A fused single launch per call -- prologue and team phases of all
rounds in one 1024-thread workgroup per stream -- was tried: 3011 frames/s against 3385 for the launch pair per round;
the merged kernel spills and its sixteen waves make the round barriers slower than the kernel boundaries were.)"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ii-vision_amd", "transcoder")]
import torch
import _iiv_native as native, stream_batch, palette
for mode in (native.DHGR, native.HGR):
    _, dm = native.cie2000_matrix(palette.NTSCPalette.rgb_array())
    table = native.build_table(mode, dm, True); store = native.build_store_table(mode, dm)
    for n in [int(x) for x in (sys.argv[1:] or ("1", "64", "512"))]:
        fm, fa = stream_batch.synth_frames_torch(n, 260, mode == native.DHGR, seed=99)
        for kern in (True, "team"):
            b = stream_batch.StreamBatch(mode, table, store, n, seeds=[(i + 1, i + 1) for i in range(n)], dm=dm)
            b.enc.set_greedy_kernel(kern)
            b.encode_frames(fm, fa, 10)
            segs = b.clock.segments(250)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            b.enc.encode(fm, fa, segs)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            b.enc.check()
            print("mode", mode, "streams", n, "kernel", kern, "fps/stream %.0f" % (250 / (t2 - t0)),
                  "total fps %.0f" % (n * 250 / (t2 - t0)), "(host submit %.1f ms of %.1f ms)" % (1e3 * (t1 - t0), 1e3 * (t2 - t0)))
            b.close()
