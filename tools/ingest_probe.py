"""Probe: iiv_frames_to_memory_maps alone -- frames/s of the ordered-dither and error-diffusion kernels on picture-like
synthetic RGB (stream_batch.synth_rgb_torch), for a few frame counts.
    python tools/ingest_probe.py [frames per call] [mode: DHGR|HGR]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ii-vision_amd", "transcoder"))
import numpy as np, torch
import _iiv_native as native, stream_batch, palette

N = int(sys.argv[1]) if len(sys.argv) > 1 else 102400
mode = native.HGR if (len(sys.argv) > 2 and sys.argv[2] == "HGR") else native.DHGR
pal = palette.NTSCPalette.rgb_array()
clips, frames = 256, N // 256
if os.environ.get("IIV_PROBE_NOISE") == "1":   # (one kernel instead of thousands: for counter runs)
    rgb = torch.randint(0, 256, (clips * frames, 192, 280, 3), dtype=torch.uint8, device="cuda")
else:
    rgb = stream_batch.synth_rgb_torch(clips, frames, seed=3).view(-1, 192, 280, 3)
main = torch.empty((rgb.shape[0], 32, 256), dtype=torch.uint8, device="cuda")
aux = torch.empty_like(main)
for name, d in (("none", 0), ("ordered 32", 32), ("diffusion", native.DITHER_DIFFUSION)):
    native.frames_to_memory_maps(mode, pal, rgb, d, out=(main, aux))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    R = 3
    for _ in range(R):
        native.frames_to_memory_maps(mode, pal, rgb, d, out=(main, aux))
    e1.record()
    torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) * 1e-3 / R
    print("%-12s %8d frames  %.3f ms  %.2f M frames/s  %.0f GB/s" % (name, rgb.shape[0], dt * 1e3, rgb.shape[0] / dt / 1e6,
                                                                     rgb.shape[0] * (161280 + (16384 if mode == native.DHGR else 8192)) / dt / 1e9), flush=True)
