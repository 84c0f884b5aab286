"""Probe: do the prologue (VALU/LDS-latency-bound) and greedy (L1-gather-bound) kernels
overlap when two halves of the clips are driven from two HIP streams?

    python tools/overlap_probe.py [streams_total] [offset]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ii-vision_amd", "transcoder")]
import torch  # noqa: E402
import _iiv_native as native  # noqa: E402
import palette  # noqa: E402
import stream_batch  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
OFFSET = int(sys.argv[2]) if len(sys.argv) > 2 else 0
G = int(sys.argv[3]) if len(sys.argv) > 3 else 2
F, STEPS = 50, 3
mode = native.DHGR
_, dm = native.cie2000_matrix(palette.NTSCPalette.rgb_array())
table = native.build_table(mode, dm, True)
store = native.build_store_table(mode, dm)
n_frames = (STEPS + 1) * F + 4


def make(n, seed):
    fm, fa = stream_batch.synth_frames_torch(n, n_frames, True, seed=seed)
    b = stream_batch.StreamBatch(mode, table, store, n, seeds=[(i, i + 7) for i in range(n)], dm=dm)
    ops = torch.empty((n, F * 490, 6), dtype=torch.uint8, device="cuda")
    return b, fm, fa, ops


ref = make(S, 1)
ref[0].encode_frames(ref[1], ref[2], F, ref[3])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(STEPS):
    ref[0].encode_frames(ref[1], ref[2], F, ref[3])
torch.cuda.synchronize()
t_ref = time.perf_counter() - t0
print("one encoder, %d streams: %.1f ms/step  %.0f fps" % (S, 1e3 * t_ref / STEPS, STEPS * F * S / t_ref))
ref[0].close()
del ref

parts = [make(S // G, 10 + g) for g in range(G)]
streams = [torch.cuda.Stream() for _ in range(G)]
for g, (p, st) in enumerate(zip(parts, streams)):
    with torch.cuda.stream(st):
        p[0].encode_frames(p[1], p[2], F, p[3])
        if OFFSET and g:   # put the groups out of phase
            p[0].encode_frames(p[1], p[2], 1, p[3])
torch.cuda.synchronize()
t0 = time.perf_counter()
# frame-granular submission, round-robin, so that the queues stay interleaved
for _ in range(STEPS):
    for f in range(F):
        for p, st in zip(parts, streams):
            with torch.cuda.stream(st):
                p[0].encode_frames(p[1], p[2], 1, p[3])
torch.cuda.synchronize()
t_par = time.perf_counter() - t0
print("%d encoders x %d streams on %d HIP streams (offset %d): %.1f ms/step  %.0f fps  (x%.3f)" % (
    G, S // G, G, OFFSET, 1e3 * t_par / STEPS, STEPS * F * S / t_par, t_ref / t_par))
for p in parts:
    p[0].enc.check()
