"""Where do bench.py's e2e legs lose 45 ms per step on ordered-dither frames?  The leg's own loop (encode -> emit_chunk ->
D2H copy to pinned memory, double-buffered, a new 50-frame scene per step) with the pieces switched on one by one, timed per
step and per kernel class (iiv_encoder_profile: HIP events around every launch).   python tools/e2e_copy_probe.py   (GPU)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ii-vision_amd", "transcoder"))
import numpy as np
import torch
import _iiv_native as native
import palette, stream_batch as sb

S, F, K = 14336, 50, 4
mode = native.DHGR
pal = palette.NTSCPalette
_, dm = native.cie2000_matrix(pal.rgb_array())
table = native.build_table(mode, dm, True)
store = native.build_store_table(mode, dm)
distinct = 512
rgb = sb.synth_rgb_torch(distinct, K * F, seed=10).view(distinct, K, F, 192, 280, 3).transpose(0, 1).contiguous()
pre = []
for k in range(K):
    m = torch.empty((S, F, 32, 256), dtype=torch.uint8, device="cuda")
    a = torch.empty((S, F, 32, 256), dtype=torch.uint8, device="cuda")
    src = rgb[k].view(distinct * F, 192, 280, 3)
    for s0 in range(0, S, distinct):
        native.frames_to_memory_maps(mode, pal.rgb_array(), src, 32, out=(m[s0:s0 + distinct], a[s0:s0 + distinct]))
    pre.append((m, a))
del rgb
torch.cuda.synchronize()
rng = np.random.default_rng(0)
tick_addr = torch.from_numpy(rng.integers(0x4000, 0x7fff, 1024).astype(np.int16)).cuda()
n_ops = F * 490
ops = torch.empty((S, n_ops, 6), dtype=torch.uint8, device="cuda")
width = native.emit_chunk_range(mode, 0, n_ops)[1] + 16
dev = [torch.empty(S * width, dtype=torch.uint8, device="cuda") for _ in range(2)]
host = [torch.empty(S * width, dtype=torch.uint8, pin_memory=True) for _ in range(2)]

for emit, copy, steps in ((False, False, 4), (True, False, 4), (True, True, 4), (True, True, 12)):
    b = sb.StreamBatch(mode, table, store, S, seeds=[(i + 1, i + 7) for i in range(S)], dm=dm)
    copy_stream = torch.cuda.Stream()
    done, ready = [torch.cuda.Event(), torch.cuda.Event()], [torch.cuda.Event(), torch.cuda.Event()]
    first_op = 0
    main = torch.cuda.current_stream()
    for ev in done:
        ev.record()
    marks = []

    def step(k):
        global first_op
        j = k & 1
        m, a = pre[k % K]
        view, segs = b.encode_frames(m, a, F, ops, loop=True)
        n = sum(s_[3] for s_ in segs)
        if emit:
            main.wait_event(done[j])
            nbytes = native.emit_chunk_range(mode, first_op, n)[1]
            o = dev[j][: S * nbytes].view(S, nbytes)
            native.emit_chunk(mode, view, first_op, tick_addr, 0xBA72, o)
            ready[j].record()
            if copy:
                with torch.cuda.stream(copy_stream):
                    copy_stream.wait_event(ready[j])
                    host[j][: S * nbytes].copy_(dev[j][: S * nbytes], non_blocking=True)
                    done[j].record()
        first_op += n
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append(e)

    step(0)
    torch.cuda.synchronize()
    b.enc.profile(True)
    marks.clear()
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    t0 = time.perf_counter()
    for k in range(1, 1 + steps):
        step(k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    p = b.enc.profile_read()
    per = [round(x.elapsed_time(y), 1) for x, y in zip([e0] + marks[:-1], marks)]
    print("emit %-5s copy %-5s %2d steps: %.1f ms per step (wall, incl. the last copy); on the launch stream per step: %s; prologue %.3f ms x %d, greedy %.3f ms x %d per step"
          % (emit, copy, steps, 1e3 * dt / steps, per, p["prologue_ms"] / p["prologue_launches"], p["prologue_launches"] // steps, p["greedy_ms"] / p["greedy_launches"], p["greedy_launches"] // steps))
    b.close()
