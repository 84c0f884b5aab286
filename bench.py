#!/usr/bin/env python3
"""DHGR frames transcoded per second on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--streams S] [--frames-per-step F]

Workload (BASELINE.json configs[3]): DHGR, NTSC palette, synthetic
560x192 S-iid clips, driver = movie.Movie.encode control flow without audio (490
opcodes per 30 fps frame, bank flip every 2 KiB of output).  One video is a strictly
sequential chain, so a GPU is filled with S independent clips (one workgroup each,
no exchange between them); N GPUs run N*S clips with no collective on the data path
("weak" scaling).  A step = --frames-per-step consecutive frames of every clip; the
defaults (20 steps x 50 frames) make each clip 1000 frames long.

Inputs (targets, tables, stream state) are resident in HBM before the timed
region.  The timed region is exactly K steps, bracketed by barrier +
torch.cuda.synchronize() on both sides; the reported time is the MAX over ranks.

Extra objects on the JSON line:
  roofline      greedy_kernel (dominant): algorithmic bytes per launch (534 B per
                opcode) / mean launch duration from HIP events
                recorded on the launch stream, against the 8 TB/s HBM peak.
  cpu_baseline  the oracle (C port of the reference path, single thread) timed on
                this host on a bounded sample of the same workload.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "ii-vision_amd", "transcoder")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

OPS_PER_FRAME = 490                # 14700 Hz / 30 fps (video.py:31-33)
BYTES_PER_OPCODE = 534             # 256 x 2 B gathers + 6 B out + 2 x 8 B packed RMW
BYTES_PER_PROLOGUE = 147456        # 2 x 32 KiB packed + 16 KiB gathers + 64 KiB priority r/w
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: 8 TB/s spec
HBM_MEASURED_READ_GBS = 5990.0     # tools/hbm_copy_bench.py on the same box (profiles/r01f_hbm_copy.txt); copy 4610, write 6900
GATHER_CEILING_LINES_PER_S = 265e9  # tools/gather_bench.hip, MI355X: 0.43 distinct lines / CU / cycle


def rank_seeds(rank, streams):
    """(random.seed, np.random.seed) of every stream owned by `rank`: disjoint across ranks."""
    return [(1 + s + streams * rank, 1 + s + streams * rank) for s in range(streams)]


def data_seed(rank):
    return 7 + 1000 * rank


def max_over_ranks(elapsed, device, world):
    """The only data that crosses ranks: MAX of the timed region."""
    if world <= 1:
        return elapsed
    import torch
    import torch.distributed as dist
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("IIV_BENCH_STREAMS", "0")),
                    help="independent clips per GPU (0 = the largest of 12288 / 6144 / 3072 / 1536 whose clips fit "
                         "the free HBM: 6144 fill the GPU at 24 per CU, twice as many hide the tail of a launch)")
    ap.add_argument("--frames-per-step", type=int, default=50)
    ap.add_argument("--mode", choices=["DHGR", "HGR"], default="DHGR")
    ap.add_argument("--coherent", action="store_true", help="S-coh input instead of S-iid")
    ap.add_argument("--img", action="store_true", help="S-img input (dithered moving bars) instead of S-iid")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=600, help="frames of stream 0 the CPU baseline encodes")
    ap.add_argument("--cpu-frames-all", type=int, default=60, help="frames per stream of the all-cores CPU baseline")
    ap.add_argument("--dw-table", action="store_true",
                    help="gather diff weights from the HBM table instead of recomputing them (same values)")
    ap.add_argument("--greedy", choices=["auto", "wave", "workgroup"], default="auto",
                    help="greedy kernel shape: one wave per stream, one 256-thread workgroup per stream, or auto")
    ap.add_argument("--full-sort", action="store_true", help="disable the prologue's prefix sort")
    ap.add_argument("--lds-pad", type=int, default=-1, help="tuning: extra LDS bytes per greedy wave (caps streams per CU)")
    ap.add_argument("--single-stream", action="store_true", help="also time one clip alone (latency-bound rate)")
    return ap.parse_args()


def main():
    args = parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    n_gpus = max(world, 1)

    import _iiv_native as native
    import stream_batch

    mode = native.DHGR if args.mode == "DHGR" else native.HGR
    dhgr = mode == native.DHGR
    F = args.frames_per_step
    total_steps = args.warmup + args.steps
    n_frames = total_steps * F
    S = args.streams
    if S <= 0:
        free_b, _ = torch.cuda.mem_get_info()
        per_clip = n_frames * 8192 * (2 if dhgr else 1) + 260 * 1024 + F * 490 * 6   # frames + stream state + opcodes
        S = next((c for c in (12288, 6144, 3072, 1536) if c * per_clip + (3 << 30) <= 0.85 * free_b), 1536)
        if world > 1:   # every rank runs the same number of clips
            t = torch.tensor([S], dtype=torch.int64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            S = int(t.item())

    # ---- setup (untimed): tables, synthetic clips, stream state, all in HBM
    t_tab = time.time()
    import palette
    _, dm = native.cie2000_matrix(palette.NTSCPalette.rgb_array())
    table = native.build_table(mode, dm, True)
    store = native.build_store_table(mode, dm)
    torch.cuda.synchronize()
    t_tab = time.time() - t_tab
    if args.img:
        fm, fa = stream_batch.synth_frames_img(S, n_frames, dhgr, seed=data_seed(rank))
    else:
        fm, fa = stream_batch.synth_frames_torch(S, n_frames, dhgr, seed=data_seed(rank), coherent=args.coherent)
    seeds = rank_seeds(rank, S)
    batch = stream_batch.StreamBatch(mode, table, store, S, seeds=seeds, dm=dm)
    if args.dw_table:
        batch.enc.set_diff_weights_mode(False)
    batch.enc.set_greedy_kernel(None if args.greedy == "auto" else args.greedy == "wave")
    if args.full_sort:
        batch.enc.set_prefix_sort(False)
    if args.lds_pad >= 0:
        batch.enc.set_greedy_lds_pad(args.lds_pad)
    ops_buf = torch.empty((S, F * OPS_PER_FRAME, 6), dtype=torch.uint8, device="cuda")

    def barrier():
        if world > 1:
            dist.barrier()

    def run_step():
        return batch.encode_frames(fm, fa, F, ops_buf)

    first_ops = None   # stream 0's opcodes of the first F frames, checked against the oracle below
    for i in range(args.warmup):
        _, segs0 = run_step()
        if i == 0:   # (streams are packed at the call's own opcode count, see iiv_encode)
            first_ops = ops_buf.view(-1)[: 6 * sum(s[3] for s in segs0)].clone().view(-1, 6)
    batch.enc.check()
    batch.enc.profile(True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    op_count, seg_count = 0, 0
    for i in range(args.steps):
        _, segs = run_step()
        if first_ops is None and i == 0:   # (only when there is no warm-up step; async D2D copy)
            first_ops = ops_buf.view(-1)[: 6 * sum(s[3] for s in segs)].clone().view(-1, 6)
        op_count += sum(s[3] for s in segs)
        seg_count += len(segs)
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    batch.enc.check()
    elapsed = t1 - t0
    prof = batch.enc.profile_read()
    batch.enc.profile(False)

    elapsed = max_over_ranks(elapsed, torch.device("cuda"), world)

    frames_done = args.steps * F * S * n_gpus
    fps = frames_done / elapsed

    out = {
        "metric": "DHGR frames transcoded/sec" if dhgr else "HGR frames transcoded/sec",
        "value": fps,
        "unit": "frames/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1000.0 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u16",
        "data": "synthetic",
        "config": {
            "workload": "%s NTSC %dx192 S-%s synthetic clips, %d frames each, %d independent clips per GPU, "
                        "Movie.encode control flow (490 opcodes/frame%s)" % (
                            args.mode, 560 if dhgr else 280, "img" if args.img else "coh" if args.coherent else "iid",
                            args.steps * F, S, ", bank flip per 2 KiB" if dhgr else ""),
            "streams_per_gpu": S,
            "frames_per_step": F,
            "opcodes_per_frame": OPS_PER_FRAME,
            "parallelism": "%d GPU x %d independent streams, no collective" % (n_gpus, S),
        },
        "opcodes_per_s": fps * OPS_PER_FRAME,
        "diff_weights": "table-gather" if args.dw_table else "recurrence",
        "table_build_s": t_tab,
    }

    if rank == 0:
        # ---- roofline of the dominant kernel, from HIP events on the launch stream
        g_ms, g_n = prof["greedy_ms"], max(prof["greedy_launches"], 1)
        p_ms, p_n = prof["prologue_ms"], max(prof["prologue_launches"], 1)
        greedy_bytes = float(op_count) * S * BYTES_PER_OPCODE       # all launches of the timed region
        achieved = greedy_bytes / (g_ms * 1e-3) / 1e9 if g_ms > 0 else 0.0
        out["roofline"] = {
            "kernel": "greedy_wave_kernel" if (args.greedy == "wave" or (args.greedy == "auto" and S >= 1536)) else "greedy_kernel",
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "peak_measured_read": HBM_MEASURED_READ_GBS,
            "traffic": _pmc_traffic(),
            "algorithmic_bytes_per_launch": greedy_bytes / g_n,
            "avg_launch_ms": g_ms / g_n,
            "launches": prof["greedy_launches"],
            # what actually bounds this kernel: 256 random store-table lookups per opcode through the
            # CU's L1 (TCP).  Ceiling = tools/gather_bench.hip on this GPU model, L2-resident table,
            # fully divergent wave64 loads (profiles/*gather_bench.txt): distinct lines per second.
            "gather": {
                "lookups_per_s": float(op_count) * S * 256 / (g_ms * 1e-3) if g_ms > 0 else 0.0,
                "ceiling_divergent_lines_per_s": GATHER_CEILING_LINES_PER_S,
                "note": "lookups that share a 128 B line inside one load instruction count once against the ceiling",
            },
        }
        pro_bytes = float(seg_count) * S * BYTES_PER_PROLOGUE
        out["roofline_prologue"] = {
            "kernel": "prologue_kernel",
            "bound": "hbm",
            "achieved": pro_bytes / (p_ms * 1e-3) / 1e9 if p_ms > 0 else 0.0,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "avg_launch_ms": p_ms / p_n,
            "launches": prof["prologue_launches"],
        }
        out["kernel_time_share"] = {"greedy": g_ms / (1000 * elapsed), "prologue": p_ms / (1000 * elapsed)}

        if args.single_stream:
            out["single_stream"] = _single_stream(native, stream_batch, mode, table, store, dm, dhgr, args)

        if not args.no_cpu_baseline and n_gpus == 1:
            out["cpu_baseline"] = _cpu_baseline(mode, dhgr, fm, fa, seeds[0], args,
                                                ops_check=(first_ops.cpu().numpy(), F))
            out["cpu_baseline_all_cores"] = _cpu_baseline_all_cores(mode, dhgr, fm, fa, seeds, args)

        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def _pmc_traffic():
    """HBM bytes per greedy_kernel launch from the committed rocprofv3 PMC summary
    (profiles/pmc_latest.json), or None when no counter run has been recorded."""
    p = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(p) as f:
            return json.load(f).get("greedy_kernel_hbm_bytes_per_launch")
    except Exception:
        return None


def _single_stream(native, stream_batch, mode, table, store, dm, dhgr, args):
    import torch
    fm, fa = stream_batch.synth_frames_torch(1, 60, dhgr, seed=99, coherent=args.coherent)
    b = stream_batch.StreamBatch(mode, table, store, 1, seeds=[(1, 1)], dm=dm)
    b.enc.set_greedy_kernel(None if args.greedy == "auto" else args.greedy == "wave")
    b.encode_frames(fm, fa, 10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b.encode_frames(fm, fa, 50)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    b.close()
    return {"value": 50 / dt, "unit": "frames/s", "us_per_opcode": 1e6 * dt / (50 * OPS_PER_FRAME)}


def _cpu_baseline(mode, dhgr, fm, fa, seed, args, ops_check):
    """Oracle (single-thread C port of video.py/screen.py) on stream 0's first frames."""
    import numpy as np
    import oracle as O
    import stream_batch
    n = min(args.cpu_frames, fm.shape[1])
    main = fm[0, :n].cpu().numpy()
    aux = fa[0, :n].cpu().numpy() if fa is not None else None
    _, dm = O.cie2000_matrix(O.PALETTE_RGB[5])
    tab = O.build_table(mode, dm, symmetric=True)   # untimed, like the GPU's table build
    v = O.Video(mode, tab, seed_py=seed[0], seed_np=seed[1])
    segs = stream_batch.MovieClock(dhgr).segments(n)
    t0 = time.perf_counter()
    got = []
    for (fr, ia, _, k) in segs:
        v.encode_frame(main[fr], aux[fr] if aux is not None else None, ia)
        got.append(v.next(k))
    dt = time.perf_counter() - t0
    parity = None
    if ops_check is not None:
        # the oracle as the checker: the GPU's opcode stream of this clip's first frames, bit for bit
        gpu_ops, nf = ops_check
        cpu_ops = np.concatenate(got)[: gpu_ops.shape[0]]
        parity = {"stream": 0, "frames": int(min(nf, n)), "opcodes": int(cpu_ops.shape[0]),
                  "equal": bool(cpu_ops.shape == gpu_ops[: cpu_ops.shape[0]].shape
                                and (cpu_ops == gpu_ops[: cpu_ops.shape[0]]).all())}
    return {
        "parity_vs_oracle": parity,
        "value": n / dt,
        "unit": "frames/s",
        "cores": 1,
        "kind": "port",
        "sample": "stream 0, first %d frames of the same clip, oracle/iiv_oracle.c (heap form), 1 thread on %s (%d cpus)"
                  % (n, _cpu_model(), os.cpu_count()),
        "seconds": dt,
    }


def _cpu_baseline_all_cores(mode, dhgr, fm, fa, seeds, args):
    """One independent stream per host thread (the oracle's C calls release the GIL)."""
    import concurrent.futures
    import numpy as np
    import oracle as O
    import stream_batch
    threads = min(os.cpu_count() or 1, fm.shape[0])
    n = min(args.cpu_frames_all, fm.shape[1])
    main = fm[:threads, :n].cpu().numpy()
    aux = fa[:threads, :n].cpu().numpy() if fa is not None else None
    _, dm = O.cie2000_matrix(O.PALETTE_RGB[5])
    tab = O.build_table(mode, dm, symmetric=True)
    segs = stream_batch.MovieClock(dhgr).segments(n)
    vids = [O.Video(mode, tab, seed_py=seeds[i][0], seed_np=seeds[i][1]) for i in range(threads)]

    def work(i):
        v = vids[i]
        for (fr, ia, _, k) in segs:
            v.encode_frame(main[i, fr], aux[i, fr] if aux is not None else None, ia)
            v.next(k)

    t0 = time.perf_counter()
    with concurrent.futures.ThreadPoolExecutor(threads) as ex:
        list(ex.map(work, range(threads)))
    dt = time.perf_counter() - t0
    return {
        "value": threads * n / dt,
        "unit": "frames/s",
        "cores": threads,
        "kind": "port",
        "sample": "%d streams x first %d frames, one oracle instance per thread on %s (%d cpus)"
                  % (threads, n, _cpu_model(), os.cpu_count()),
        "seconds": dt,
    }


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
