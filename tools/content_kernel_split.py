"""Per-launch prologue / greedy time by CONTENT: S-iid, S-img (converged picture-like clips), and frames converted from the
synthetic RGB clips of bench.py's e2e legs (ordered dither, error diffusion; a new scene every 50 frames) -- the kernels'
own HIP-event timing (iiv_encoder_profile).   python tools/content_kernel_split.py [streams]   (needs the GPU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ii-vision_amd", "transcoder"))
import numpy as np
import torch
import _iiv_native as native
import palette, stream_batch as sb

S = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 14336
FORM = next((a for a in sys.argv[1:] if a in ("shared", "plain")), None)    # force the one-wave kernel's form (default: the encoder picks)
F = 50
mode = native.HGR if "HGR" in sys.argv[1:] else native.DHGR
DH = mode == native.DHGR
pal = palette.NTSCPalette
_, dm = native.cie2000_matrix(pal.rgb_array())
table = native.build_table(mode, dm, True)
store = native.build_store_table(mode, dm)


def run(name, fm, fa, steps=3):
    b = sb.StreamBatch(mode, table, store, S, seeds=[(i + 1, i + 7) for i in range(S)], dm=dm)
    if FORM:
        b.enc.set_greedy_kernel(FORM)
    ops = torch.empty((S, F * 490, 6), dtype=torch.uint8, device="cuda")
    b.encode_frames(fm, fa, F, ops, loop=True)      # warm-up: the first 50 frames from an empty screen
    b.enc.check()
    b.enc.profile(True)
    for _ in range(steps):
        b.encode_frames(fm, fa, F, ops, loop=True)
    b.enc.check()
    p = b.enc.profile_read()
    st = b.enc.input_stats()
    print("%-6s %-34s prologue %.4f ms  greedy %.4f ms per launch (%d launches); nonce-decided share %.3f, form %s" % (
        FORM or "auto", name, p["prologue_ms"] / max(p["prologue_launches"], 1), p["greedy_ms"] / max(p["greedy_launches"], 1), p["greedy_launches"], st[0], st[1]))
    b.close()


fm, fa = sb.synth_frames_torch(S, 4 * F, DH, seed=5)
run("S-iid", fm, fa)
del fm, fa
torch.cuda.empty_cache()
fm, fa = sb.synth_frames_img(2048, 4 * F, DH, seed=5)
reps = (S + 2047) // 2048
run("S-img (2048 distinct, tiled)", fm.repeat(reps, 1, 1, 1)[:S].contiguous(), fa.repeat(reps, 1, 1, 1)[:S].contiguous() if DH else None)
del fm, fa
torch.cuda.empty_cache()
distinct = 512
rgb = sb.synth_rgb_torch(distinct, 4 * F, seed=10)
for name, dither in (("RGB, ordered dither 32", 32), ("RGB, error diffusion", native.DITHER_DIFFUSION)):
    m = torch.empty((distinct, 4 * F, 32, 256), dtype=torch.uint8, device="cuda")
    a = torch.empty((distinct, 4 * F, 32, 256), dtype=torch.uint8, device="cuda") if DH else None
    native.frames_to_memory_maps(mode, pal.rgb_array(), rgb.view(distinct * 4 * F, 192, 280, 3), dither, out=(m, a))
    torch.cuda.synchronize()
    reps = (S + distinct - 1) // distinct
    run(name + " (512 distinct, tiled)", m.repeat(reps, 1, 1, 1)[:S].contiguous(), a.repeat(reps, 1, 1, 1)[:S].contiguous() if DH else None)
    del m, a
    torch.cuda.empty_cache()
