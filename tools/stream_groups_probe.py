"""Probe: does splitting a batch over G encoders, each on its own HIP stream, raise the total rate?
One stream: prologue(all) -> greedy(all) -> ... in order; every launch ends with a tail in which the CUs
run partly empty (streams differ by +-20 % in how long a launch takes them).  With G groups on G HIP
streams one group's tail overlaps another group's next launch.
    python tools/stream_groups_probe.py [total streams] [frames per step] [groups ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ii-vision_amd", "transcoder"))
import numpy as np, torch
import _iiv_native as native, stream_batch, palette

S = int(sys.argv[1]) if len(sys.argv) > 1 else 14336
F = int(sys.argv[2]) if len(sys.argv) > 2 else 50
groups = [int(a) for a in sys.argv[3:]] or [1, 2, 4]
KIND = os.environ.get("IIV_PROBE_KIND", "iid")
_, dm = native.cie2000_matrix(palette.NTSCPalette.rgb_array())
mode = native.DHGR
table = native.build_table(mode, dm, True)
store = native.build_store_table(mode, dm)
n_clip = 6
for G in groups:
    per = S // G
    batches, frames, streams = [], [], []
    for g in range(G):
        if KIND == "img":
            fm, fa = stream_batch.synth_frames_img(per, n_clip, True, seed=5 + g)
        else:
            fm, fa = stream_batch.synth_frames_torch(per, n_clip, True, seed=5 + g)
        b = stream_batch.StreamBatch(mode, table, store, per, seeds=[(g * per + i + 1, g * per + i + 1) for i in range(per)], dm=dm)
        b.enc.set_greedy_kernel(os.environ.get("IIV_PROBE_GREEDY") or True)
        batches.append(b)
        frames.append((fm, fa))
        streams.append(torch.cuda.Stream())
    torch.cuda.synchronize()

    def step():
        for g in range(G):
            with torch.cuda.stream(streams[g]):
                batches[g].encode_frames(frames[g][0], frames[g][1], F, loop=True)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 4
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for b in batches:
        b.enc.check()
    print("G=%d groups x %d streams: %.0f frames/s (%.1f ms per %d-frame step)" % (G, per, per * G * F * K / dt, dt / K * 1e3, F), flush=True)
    for b in batches:
        b.close()
    del batches, frames
    torch.cuda.empty_cache()
