"""f4, VERDICT r5 #8: would pruning the joint step's bytes pay?  The oracle (CPU; oracle/iiv_oracle.c: joint_prune_stats) runs
the joint content choice as movie.py paces it on S-iid / S-coh / S-img clips and counts, per step, the eligible bytes of the
page, those a descending-diff-weight walk has to look at before its bound stops it (every byte value's second smallest delta
already <= -dw of everything left), and what a kernel-friendly two-pass form could skip.
    python tools/joint_prune_rate.py [frames]  > profiles/r06_joint_prune_rate.txt      (CPU only, ~2 min)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "ii-vision_amd", "transcoder"))
import oracle as O  # noqa: E402
import stream_batch  # noqa: E402

O.build()
n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dm = O.cie2000_matrix(O.PALETTE_RGB[5])[1]
print("# the joint step's prune rate (tools/joint_prune_rate.py; oracle, NTSC palette, movie.py pacing, %d frames per clip)" % n_frames)
print("# per step: eligible = bytes of the page with priority != 0 besides the primary; looked at = bytes a walk in descending dw")
print("# scores before every byte value's k-th smallest delta is <= -dw of all that is left (k = 2; 3 with the fourth offset);")
print("# two-pass = share of the bytes behind the 16 largest that the bound reached after those 16 would let a second pass skip")
for mode, mname in ((1, "DHGR"), (0, "HGR")):
    table = O.build_table(mode, dm, symmetric=True)
    for kind in ("iid", "coh", "img"):
        for fourth in (False, True):
            if kind == "img":
                fm, fa = stream_batch.synth_frames_img(1, n_frames, bool(mode), seed=5, device="cpu")
            else:
                fm, fa = stream_batch.synth_frames_torch(1, n_frames, bool(mode), seed=5, coherent=(kind == "coh"), device="cpu")
            fm = fm[0].numpy()
            fa = fa[0].numpy() if fa is not None else None
            v = O.Video(mode, table, seed_py=1, seed_np=2)
            v.set_joint(True)
            v.set_fourth_offset(fourth)
            v.joint_stats(True)
            prev = None
            n_ops = 0
            for (f, a, _, k) in stream_batch.MovieClock(bool(mode)).segments(n_frames):
                if f != prev:
                    v.reset_out_of_work()
                    prev = f
                v.encode_frame(fm[f], fa[f] if mode else None, int(a))
                v.next(k)
                n_ops += k
            st = v.joint_stats(True)
            s = max(st["steps"], 1)
            print("%-4s S-%-3s %s: %6d joint steps of %6d opcodes; eligible %6.1f per step (dw == 0: %4.1f); looked at %6.1f (%.1f %% pruned); two-pass: %.1f %% of the %5.1f behind the first 16"
                  % (mname, kind, "fourth" if fourth else "      ", st["steps"], n_ops, st["eligible"] / s, st["eligible_dw0"] / s, st["looked_at"] / s,
                     100.0 * (1.0 - st["looked_at"] / max(st["eligible"], 1)), 100.0 * st["behind_16_prunable"] / max(st["behind_16"], 1), st["behind_16"] / s))
            sys.stdout.flush()
