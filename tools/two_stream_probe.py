"""Experiment: one batch of S clips on one HIP stream against two half-batches on two HIP streams (the kernels of one half
beside the other half's: a launch's tail -- its last streams finishing -- filled by the other half's next kernel).
    python tools/two_stream_probe.py [S] [steps] [DHGR|HGR]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ii-vision_amd", "transcoder"))
import numpy as np, torch
import _iiv_native as native, stream_batch, palette

S = int(sys.argv[1]) if len(sys.argv) > 1 else 14336
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
mode = native.HGR if (len(sys.argv) > 3 and sys.argv[3] == "HGR") else native.DHGR
dhgr = mode == native.DHGR
F = 50
_, dm = native.cie2000_matrix(palette.NTSCPalette.rgb_array())
table = native.build_table(mode, dm, True); store = native.build_store_table(mode, dm)
n_frames = steps * F


def run(parts):
    per = S // parts
    batches, clips, streams, bufs = [], [], [], []
    for p in range(parts):
        fm, fa = stream_batch.synth_frames_torch(per, n_frames, dhgr, seed=11 + p)
        clips.append((fm, fa))
        batches.append(stream_batch.StreamBatch(mode, table, store, per, seeds=[(p * per + i + 1, p * per + i + 1) for i in range(per)], dm=dm))
        streams.append(torch.cuda.Stream() if parts > 1 else torch.cuda.current_stream())
        bufs.append(torch.empty((per, F * 490, 6), dtype=torch.uint8, device="cuda"))
    torch.cuda.empty_cache()

    def step():
        for p in range(parts):
            with torch.cuda.stream(streams[p]):
                batches[p].encode_frames(clips[p][0], clips[p][1], F, bufs[p], loop=True)
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for b in batches:
        b.enc.check()
        b.close()
    return steps * F * per * parts / dt


for parts in (1, 2, 1, 2):
    print("%d stream(s) x %d clips: %.0f frames/s" % (parts, S // parts, run(parts)), flush=True)
    torch.cuda.empty_cache()
