#!/usr/bin/env python3
"""ratio of the C port (oracle) to the REFERENCE's own Python,
both timed in the build container on the same synthetic workload (DHGR / HGR, NTSC, S-iid,
Movie.encode control flow without audio, 490 opcodes per frame).  The reference cannot travel
to the GPU box, so bench.py multiplies (GPU fps / port fps on the box) by this ratio to state
"x reference" honestly.  Writes profiles/reference_ratio.json (a committed constant with its
provenance).  Needs /root/reference; run:  python tools/measure_reference_ratio.py
"""
import contextlib
import io
import json
import os
import platform
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "ii-vision_amd", "transcoder"))


def main():
    import oracle as O
    import make_golden as G
    O.build()
    G.setup_reference("/tmp/iiv_ref")
    G.write_reference_tables("/tmp/iiv_ref", O)
    import frame_grabber
    import palette
    import screen
    import video
    import video_mode
    out = {"host": platform.processor() or platform.machine(), "cpus": os.cpu_count(),
           "model": next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?"),
           "numpy": np.__version__, "python": platform.python_version(),
           "workload": "NTSC, S-iid frames (default_rng(7)), random.seed(1) / np.random.seed(1), Movie pacing "
                       "(490 opcodes per frame, DHGR bank flips), 5 warm-up + 30 timed frames, 1 thread"}
    for mode_name in ("DHGR", "HGR"):
        mode = video_mode.VideoMode[mode_name]
        nf = 35
        frames = G.synth_frames(mode_name, nf, 7)
        sched = G.movie_schedule(mode_name, nf)
        warm = [s for s in sched if s[0] < 5]
        timed = [s for s in sched if s[0] >= 5]
        # ---- the reference
        random.seed(1)
        np.random.seed(1)
        pal = palette.Palette.NTSC
        v = video.Video(frame_grabber.FrameGrabber(mode), ticks_per_second=14700., mode=mode, palette=pal)

        def run_ref(part):
            n = 0
            for (fi, is_aux, n_ops) in part:
                main_mm = screen.MemoryMap(screen_page=1, page_offset=frames[fi, 0].copy())
                if mode_name == "DHGR":
                    aux_mm = screen.MemoryMap(screen_page=1, page_offset=frames[fi, 1].copy())
                    tgt = screen.DHGRBitmap(main_memory=main_mm, aux_memory=aux_mm, palette=pal)
                else:
                    tgt = screen.HGRBitmap(main_memory=main_mm, palette=pal)
                gen = v.encode_frame(tgt, is_aux=bool(is_aux))
                for _ in range(n_ops):
                    next(gen)
                n += n_ops
            return n
        with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
            run_ref(warm)
            t0 = time.perf_counter()
            n_ops = run_ref(timed)
            dt_ref = time.perf_counter() - t0
        # ---- the C port on the same frames / schedule / seeds
        _, dm = O.cie2000_matrix(O.PALETTE_RGB[5])
        tab = O.build_table(1 if mode_name == "DHGR" else 0, dm, symmetric=True)
        ov = O.Video(1 if mode_name == "DHGR" else 0, tab, seed_py=1, seed_np=1)

        def run_port(part):
            for (fi, is_aux, k) in part:
                ov.encode_frame(frames[fi, 0], frames[fi, 1] if mode_name == "DHGR" else None, is_aux)
                ov.next(k)
        run_port(warm)
        t0 = time.perf_counter()
        run_port(timed)
        dt_port = time.perf_counter() - t0
        out[mode_name] = {"reference_python_frames_per_s": 30 / dt_ref, "port_frames_per_s": 30 / dt_port,
                          "port_over_reference": dt_ref / dt_port, "opcodes": n_ops}
        print(mode_name, out[mode_name], flush=True)
    out["measured"] = time.strftime("%Y-%m-%d")
    with open(os.path.join(ROOT, "profiles", "reference_ratio.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
