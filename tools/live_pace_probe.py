"""Does the live hand-over slow the team kernel?  One clip, launches of 2000 opcodes, iiv_encode against iiv_encode_live, timed
with events on the launch stream.   python tools/live_pace_probe.py   (needs the GPU)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ii-vision_amd", "transcoder"))
import numpy as np
import torch
import _iiv_native as native
import palette, screen, stream_batch

pal = palette.Palette.NTSC
tables = screen.DHGRBitmap(main_memory=screen.MemoryMap(1), aux_memory=screen.MemoryMap(1), palette=pal).edit_distances(pal)
enc = native.Encoder(native.DHGR, tables.table, tables.store, n_streams=1, dm=tables.dm)
fm, fa = stream_batch.synth_frames_torch(1, 12, True, seed=3, device="cuda")
q = enc.live_queue(0)
ops = torch.empty((1, 2048, 6), dtype=torch.uint8, device="cuda")
for name in ("encode", "live", "encode", "live"):
    ts = []
    for f in range(12):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        if name == "live":
            enc.encode_live(fm, fa, (f, f & 1, 1, 2000), ops, 0, 100 + f)
        else:
            enc.encode(fm, fa, [(f, f & 1, 1, 2000)], ops_out=ops)
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    enc.check()
    print("%-7s prologue + 2000 opcodes: median %.0f us  (%.3f us per opcode)  min %.0f max %.0f" % (name, np.median(ts), np.median(ts) / 2000, min(ts), max(ts)))

for k in (1, 50, 292, 1000):
    ts = []
    for f in range(12):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        enc.encode(fm, fa, [(f, f & 1, 1, k)], ops_out=ops)
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    print("prologue + %4d opcodes: median %.0f us  min %.0f" % (k, np.median(ts), min(ts)))
enc.check()
t0 = time.perf_counter()
for f in range(200):
    enc.encode(fm, fa, [(f % 12, f & 1, 1, 1)], ops_out=ops)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host cost of one encode() call (prologue + greedy enqueued): %.1f us" % (1e6 * (t1 - t0) / 200))
