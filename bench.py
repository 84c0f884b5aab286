#!/usr/bin/env python3
"""DHGR frames transcoded per second on MI355X (BASELINE.json metric) + make_data_tables wall-clock.

    python bench.py --gpus N --steps K --warmup W [--mode DHGR|HGR] [--palette NTSC|IIGS] [--streams S] [--emit]

Workload (BASELINE.json configs[3]; configs[2] with --mode HGR; configs[4] with --palette IIGS
under torchrun on 8 GPUs): synthetic 560x192 (280x192) S-iid clips, driver =
movie.Movie.encode control flow without audio (490 opcodes per 30 fps frame, DHGR bank flip
every 2 KiB of output).  One video is a strictly sequential chain, so a GPU is filled with S
independent clips (one wave each, no exchange between them); N GPUs run N*S clips with no
collective on the data path ("weak" scaling).  A step = --frames-per-step consecutive frames
of every clip; the defaults (20 steps x 50 frames) make each clip 1000 frames long.

Inputs (targets, tables, stream state) are resident in HBM before the timed region.  The timed
region is exactly K steps, bracketed by barrier + torch.cuda.synchronize() on both sides; the
reported time is the MAX over ranks.  Between the W warm-up steps and the timed region runs an
untimed EVENTS leg (min(K, 3) of the same steps) in which the library records a HIP event pair
around every kernel launch on the launch stream: the per-kernel launch durations of the roofline
objects come from it, and the timed region runs without any event recording.

Extra objects on the JSON line:
  roofline            greedy_wave_kernel (dominant): algorithmic bytes per launch (534 B per opcode,
                      SURVEY.md 8d) / mean launch duration from the events leg, against the 8 TB/s
                      HBM peak.  `traffic`, `issue` and roofline_prologue's counter_* figures are
                      quoted from the committed counter run profiles/pmc_latest.json ONLY when it
                      carries this library's build id (iiv_version(); `build_id` in the line) and
                      profiled the kernel instantiation that ran most launches here; otherwise they
                      are null and traffic_source says "stale: ..." (tests/test_bench_counters.py).
  cpu_baseline        the oracle (C port of the reference path, one thread) timed on this host on a
                      bounded sample of the same workload; it also checks the GPU's opcodes of clip 0.
  vs_reference_python GPU / port-on-this-box x (port / reference Python, measured in the build
                      container: profiles/reference_ratio.json).
  make_data_tables_s  wall-clock of make_data_tables.main(): both palettes, HGR + DHGR, four files in
                      the reference's format (BASELINE metric M2; README: ~90 min).
  single_stream       one clip alone (latency-bound rate).
  emit (with --emit)  the same K steps again with the opcodes turned into the .a2m byte stream on the
                      device (iiv_emit_chunk) and copied to pinned host memory, double-buffered.
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
BYTES_PER_FRAME = {"DHGR": 657e3, "HGR": 409e3}   # (1 + 490/292) calls + 490 opcodes / 1 call + 490 opcodes
GATHER_CEILING_GLOADS_HGR = 995.0    # tools/gather_ceiling D 14336 HGR (3.61 ms per launch of 490 opcodes)
GATHER_CEILING_GLOADS = 1184.2   # tools/gather_ceiling.hip, variant D (profiles/r02l_gather_ceiling.txt)
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: 8 TB/s spec
# What an MI355X box of this pool streams, best of tools/hbm_stream.hip's launch shapes (profiles/r04_hbm_stream.txt; 4 GiB,
# 16 B per lane): copy = read + write bytes per second.  (torch's own kernels, profiles/r01f_hbm_copy.txt: copy 4610, read
# 5990, fill 6900.)
HBM_MEASURED_COPY_GBS = 5560.0
HBM_MEASURED_READ_GBS = 7070.0
PALETTE_IDS = {"NTSC": 5, "IIGS": 0}   # palette.py:18-23


def rank_seeds(rank, streams):
    """(random.seed, np.random.seed) of every stream owned by `rank`: disjoint across ranks."""
    return [(1 + s + streams * rank, 1 + s + streams * rank) for s in range(streams)]


def data_seed(rank):
    return 7 + 1000 * rank


def max_over_ranks(elapsed, device, world, use_dist=None):
    """The only data that crosses ranks: MAX of the timed region."""
    if not (world > 1 if use_dist is None else use_dist):
        return elapsed
    import torch
    import torch.distributed as dist
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_ranks(elapsed, device, world, use_dist=None):
    """Every rank's timed region (seconds), rank order -- reporting only, after the timed region."""
    if not (world > 1 if use_dist is None else use_dist):
        return [elapsed]
    import torch
    import torch.distributed as dist
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("IIV_BENCH_STREAMS", "0")),
                    help="independent clips per GPU (0 = the largest of 14336 / 12288 / 6144 / 3072 / 1536 whose clips fit "
                         "the free HBM: ~7000 fill the GPU, more hide the tail of a launch)")
    ap.add_argument("--frames-per-step", type=int, default=50)
    ap.add_argument("--mode", choices=["DHGR", "HGR"], default="DHGR")
    ap.add_argument("--palette", choices=sorted(PALETTE_IDS), default="NTSC",
                    help="NTSC (main.py's default) or IIGS = the //gs RGB palette of BASELINE config 5")
    ap.add_argument("--config", type=int, choices=[3, 4, 5], default=0,
                    help="preset named after BASELINE.json's configs: 3 = HGR NTSC 280x192, 4 = DHGR NTSC 560x192 (the default "
                         "workload), 5 = DHGR with the //gs RGB palette, the 8-GPU configuration (same as --palette IIGS); sets --mode / --palette")
    ap.add_argument("--coherent", action="store_true", help="S-coh input instead of S-iid")
    ap.add_argument("--img", action="store_true", help="S-img input (dithered moving bars) instead of S-iid")
    ap.add_argument("--img-distinct", type=int, default=0, help="with --img: render this many distinct clips and tile them over the streams (0 = all distinct)")
    ap.add_argument("--static", action="store_true",
                    help="S-static input: converging content -- every frame is the previous one with 2 %% of its bytes redrawn "
                         "(with --repeat N, every drawn frame is shown N times): the work list runs dry, the re-queued bag "
                         "and the out-of-work padding are what gets timed")
    ap.add_argument("--repeat", type=int, default=1, help="with --static: show every drawn frame this many times")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip make_data_tables / single-stream (profiling runs)")
    ap.add_argument("--extras", default="all",
                    help="comma list of the extra legs to run beside the timed region (default all): single_stream, emit, "
                         "make_data_tables, ingest, dropin, hgr, img, fourth_offset, joint -- for iterating on one of them")
    ap.add_argument("--cpu-frames", type=int, default=600, help="frames of stream 0 the CPU baseline encodes")
    ap.add_argument("--cpu-frames-all", type=int, default=60, help="frames per stream of the all-cores CPU baseline")
    ap.add_argument("--dw", choices=["split", "recurrence", "table"], default="recurrence",
                    help="how the prologue obtains the diff weights (same values): the recurrence in the kernel, two "
                         "gathers from the split table, or one gather from the full table in HBM")
    ap.add_argument("--dw-table", action="store_true", help="same as --dw table")
    ap.add_argument("--joint", nargs="?", const=True, default=False, choices=[True, "split"],
                    help="SURVEY 8(f4): choose every step's content byte jointly with its extra offsets "
                         "(IIV_CONTENT_JOINT; NOT the reference's output -- not the BASELINE workload); "
                         "`--joint split`: the second implementation (IIV_CONTENT_JOINT_SPLIT)")
    ap.add_argument("--fourth", action="store_true",
                    help="SURVEY 8(f4): up to three extra offsets per opcode instead of two and a copy of the first "
                         "(IIV_OPT_FOURTH_OFFSET; NOT the reference's output -- not the BASELINE workload)")
    ap.add_argument("--greedy", choices=["auto", "wave", "workgroup", "shared", "plain"], default="auto",
                    help="greedy kernel shape: one wave per stream, one 256-thread workgroup per stream, or auto")
    ap.add_argument("--full-sort", action="store_true", help="disable the prologue's prefix sort")
    ap.add_argument("--no-stream-order", action="store_true",
                    help="A/B: launch the streams in index order instead of longest first (IIV_OPT_STREAM_ORDER 0; same bytes)")
    ap.add_argument("--lds-pad", type=int, default=-1, help="tuning: extra LDS bytes per greedy wave (caps streams per CU)")
    ap.add_argument("--emit", action="store_true",
                    help="(default unless --no-extras) also time encode -> .a2m byte emission -> pinned host "
                         "memory, end to end; kept as a flag for older command lines")
    ap.add_argument("--no-emit", action="store_true", help="skip the end-to-end emission leg")
    ap.add_argument("--dist-backend", choices=["gloo", "nccl"], default=os.environ.get("IIV_BENCH_DIST_BACKEND", "gloo"),
                    help="process group of an N > 1 run.  The path has NO collective (north_star: 'no RCCL'): the group only carries "
                         "the barriers around the timed region and three scalars (MIN clip count, MAX elapsed, per-rank figures), "
                         "which run on CPU tensors over gloo by default; 'nccl' (= RCCL) puts the same three scalars on the device")
    ap.add_argument("--backend", default=os.environ.get("IIV_BENCH_BACKEND", ""),
                    help="tests only: 'module:Class' standing in for GpuBackend (tests/bench_standin.py runs main() "
                         "without a device; the product path has no CPU fallback)")
    args = ap.parse_args(argv)
    if args.config == 3:
        args.mode, args.palette = "HGR", "NTSC"
    elif args.config == 4:
        args.mode, args.palette = "DHGR", "NTSC"
    elif args.config == 5:
        args.mode, args.palette = "DHGR", "IIGS"
    return args


def _resolve_backend(spec):
    """'module:Class' -> the class (tests/bench_standin.py); '' -> GpuBackend."""
    if not spec:
        return GpuBackend
    import importlib
    mod, _, name = spec.partition(":")
    return getattr(importlib.import_module(mod), name)


# IIV_BENCH_REHEARSE_ON_ONE_GPU=1: every rank binds GPU 0 and the process group is gloo (RCCL refuses two ranks on one
# device).  It exists to run the launcher and the whole multi-rank control flow on a 1-GPU box -- real kernels, real
# processes, the ranks sharing the one GPU -- and says so in the line; its value is not a multi-GPU measurement.
REHEARSE = os.environ.get("IIV_BENCH_REHEARSE_ON_ONE_GPU") == "1"


def visible_gpus():
    """GPUs this process could bind ranks to, counted WITHOUT the HIP runtime (the launcher below starts children, and a
    process that has touched the GPU must not): the *_VISIBLE_DEVICES lists if set, else the GPU nodes of the KFD
    topology in sysfs.  None = unknown (no refusal is based on it; a rank without a device fails by itself)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    import glob
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    n = 0
    for f in nodes:
        try:
            props = dict(l.split()[:2] for l in open(f) if len(l.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:   # (CPU nodes have none)
                n += 1
        except Exception:
            return None
    return n


def _cpulist(text):
    """'0-3,8,10-11' -> {0, 1, 2, 3, 8, 10, 11}"""
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def pin_to_gpu_numa_node(local_rank):
    """Before any GPU call: pin this rank to the CPUs of its GPU's NUMA node (one process per GPU drives ~270 launches per
    step from one host thread; on an 8-GPU node the GPUs hang off different sockets, and a rank scheduled on the far one pays
    the hop on every launch and every pinned-memory copy).  GPU `local_rank` = the local_rank-th GPU node of the KFD
    topology (through HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES if they are lists of indices); its NUMA node is
    /sys/class/drm/renderD<drm_render_minor>/device/numa_node; the CPUs are that node's cpulist cut with what this process may
    run on.  Returns what it did -- {"numa_node", "cpus", "n_cpus", "pinned"} -- and never fails the run: anything it
    cannot read leaves the affinity alone and says why.  (IIV_BENCH_SYSFS: another root for /sys, tests only.)"""
    import glob
    root = os.environ.get("IIV_BENCH_SYSFS", "/sys")
    info = {"numa_node": None, "cpus": None, "n_cpus": None, "pinned": False}
    try:
        gpus = []
        for f in sorted(glob.glob(root + "/class/kfd/kfd/topology/nodes/*/properties"), key=lambda p: int(p.split("/")[-2])):
            props = dict(l.split()[:2] for l in open(f) if len(l.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                gpus.append(int(props.get("drm_render_minor", "-1")))
        idx = local_rank
        for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
            v = os.environ.get(var)
            if v:
                ids = [x.strip() for x in v.split(",") if x.strip() != ""]
                if local_rank < len(ids) and ids[local_rank].isdigit():
                    idx = int(ids[local_rank])
                break
        if idx >= len(gpus) or gpus[idx] < 0:
            info["why"] = "no KFD GPU node %d under %s" % (idx, root)
            return info
        node = int(open("%s/class/drm/renderD%d/device/numa_node" % (root, gpus[idx])).read().strip())
        info["numa_node"] = node
        if node < 0:
            info["why"] = "the GPU reports no NUMA node"
            return info
        cpus = _cpulist(open("%s/devices/system/node/node%d/cpulist" % (root, node)).read()) & os.sched_getaffinity(0)
        if not cpus:
            info["why"] = "none of NUMA node %d's CPUs is available to this process" % node
            return info
        os.sched_setaffinity(0, cpus)
        info.update({"cpus": ",".join(str(c) for c in sorted(cpus)) if len(cpus) <= 8 else "%d..%d" % (min(cpus), max(cpus)),
                     "n_cpus": len(cpus), "pinned": True})
    except Exception as e:   # (no sysfs, no permission: run unpinned)
        info["why"] = repr(e)
    return info


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with no launcher around it: start N ranks of this script, one per GPU.

    The parent never touches a GPU (no HIP call, no libiivision.so): it picks a rendezvous port on 127.0.0.1, starts
    N children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set -- exactly what
    `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` would give them -- forwards rank 0's
    single JSON line, and exits non-zero if any rank fails (the others are then terminated by pid)."""
    import socket
    n = int(args.gpus)
    backend = _resolve_backend(args.backend)
    if getattr(backend, "is_gpu", True) and not REHEARSE:
        have = visible_gpus()
        if have is not None and have < n:
            sys.stderr.write("bench.py: --gpus %d asked for, %d GPU(s) visible to this process\n" % (n, have))
            return 2
    cmd = [sys.executable, os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv)
    # (the rendezvous port is found by binding port 0 and closing the socket: another process can take it before rank 0
    # listens on it -- a run whose ranks fail within the first minute is started once more on a fresh port)
    for attempt in (0, 1):
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        t_start = time.time()
        rc = _run_ranks(cmd, n, port)
        if rc == 0 or attempt == 1 or time.time() - t_start > 60.0:
            return rc
        sys.stderr.write("bench.py: ranks failed within %.0f s of their start; once more on a fresh rendezvous port\n" % (time.time() - t_start))
    return rc


def _run_ranks(cmd, n, port):
    import subprocess
    import tempfile
    procs = []
    with tempfile.TemporaryFile("w+") as out0:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), IIV_BENCH_LAUNCHED_BY="bench.py")
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen(cmd, env=env, stdout=out0 if r == 0 else subprocess.DEVNULL))
        # wait for all of them; the first rank that fails ends the others (by pid: they would wait for it at the
        # rendezvous or at the next barrier for ever)
        rcs = [None] * n
        while any(c is None for c in rcs):
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    rcs[r] = p.poll()
            if any(c not in (None, 0) for c in rcs):
                for r, p in enumerate(procs):
                    if rcs[r] is None:
                        p.terminate()
                        try:
                            rcs[r] = p.wait(timeout=30)
                        except subprocess.TimeoutExpired:
                            p.kill()
                            rcs[r] = p.wait()
                break
            time.sleep(0.05)
        out0.seek(0)
        line = out0.read()
    # rank 0's stdout: the JSON line goes to ours, anything else a library printed there (gloo's connection notice) to stderr
    for l in (line or "").splitlines():
        (sys.stdout if l.startswith('{"metric"') else sys.stderr).write(l + "\n")
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(rcs) if c != 0]
    if bad:
        sys.stderr.write("bench.py: ranks failed (rank, exit code): %s\n" % bad)
        return 1
    return 0


class GpuBackend:
    """Everything bench.main() does on the device.  tests/test_multiprocess_gloo.py runs main() with
    a CPU stand-in for this class, so that the multi-rank control flow is executed before hardware sees it."""

    dist_backend = "gloo"
    is_gpu = True

    def __init__(self, args, local_rank, world):
        import torch
        import _iiv_native as native
        import stream_batch
        self.torch, self.native, self.sb = torch, native, stream_batch
        if world > 1 and not REHEARSE:
            have = visible_gpus()   # (in front of set_device, which would raise first with a less helpful message)
            if have is not None and have <= local_rank:
                raise SystemExit("bench.py: rank %s has no GPU %d (%d visible)" % (os.environ.get("RANK", "?"), local_rank, have))
        torch.cuda.set_device(local_rank if world > 1 and not REHEARSE else 0)
        self.device = torch.device("cuda", torch.cuda.current_device())
        # The path has no collective: what crosses ranks is two barriers and three scalars.  They live on CPU tensors over
        # gloo unless --dist-backend nccl asks for RCCL (which two ranks sharing one device, the rehearsal, cannot use).
        self.dist_backend = "gloo" if REHEARSE else getattr(args, "dist_backend", "gloo")
        if self.dist_backend == "gloo":
            self.coll_device = torch.device("cpu")
        self.args = args
        self.mode = native.DHGR if args.mode == "DHGR" else native.HGR
        self.dhgr = self.mode == native.DHGR

    def dist_kwargs(self):
        return {"device_id": self.device} if self.dist_backend == "nccl" else {}

    def free_bytes(self):
        return self.torch.cuda.mem_get_info()[0]

    def synchronize(self):
        self.torch.cuda.synchronize()

    def build_tables(self):
        import palette
        t = time.time()
        pal = palette.PALETTES[palette.Palette(PALETTE_IDS[self.args.palette])]
        _, self.dm = self.native.cie2000_matrix(pal.rgb_array())
        self.table = self.native.build_table(self.mode, self.dm, True)
        self.store = self.native.build_store_table(self.mode, self.dm)
        self.synchronize()
        return time.time() - t

    def make_clips(self, S, n_frames, seed):
        a = self.args
        if a.img:
            # (the generator renders every frame of every clip as a (S, 192, 560) field: a bounded number of distinct clips,
            # tiled over the streams -- each stream still has its own RNG seeds, i.e. its own tie-breaking)
            distinct = min(S, getattr(a, "img_distinct", 0) or S)
            fm, fa = self.sb.synth_frames_img(distinct, n_frames, self.dhgr, seed=seed)
            if distinct < S:
                reps = -(-S // distinct)
                fm = fm.repeat((reps, 1, 1, 1))[:S].contiguous()
                fa = fa.repeat((reps, 1, 1, 1))[:S].contiguous() if fa is not None else None
            self.fm, self.fa = fm, fa
        else:
            self.fm, self.fa = self.sb.synth_frames_torch(S, n_frames, self.dhgr, seed=seed, coherent=a.coherent or a.static,
                                                          keep=0.98 if a.static else 0.9, repeat=a.repeat if a.static else 1)
        # the generators' temporaries go back to the driver: libiivision allocates with hipMalloc,
        # outside torch's caching allocator, and the clips leave it 40 GiB
        self.torch.cuda.empty_cache()

    def make_batch(self, S, seeds):
        a = self.args
        b = self.sb.StreamBatch(self.mode, self.table, self.store, S, seeds=seeds, dm=self.dm, joint_content=a.joint, fourth_offset=a.fourth)
        b.enc.set_diff_weights_mode("table" if a.dw_table else a.dw)
        b.enc.set_greedy_kernel({"auto": None, "wave": True, "workgroup": False}.get(a.greedy, a.greedy))
        if a.full_sort:
            b.enc.set_prefix_sort(False)
        if getattr(a, "no_stream_order", False):
            b.enc.set_stream_order(False)
        if a.lds_pad >= 0:
            b.enc.set_greedy_lds_pad(a.lds_pad)
        self.batch = b
        self.S = S
        self.ops_buf = self.torch.empty((S, a.frames_per_step * OPS_PER_FRAME, 6), dtype=self.torch.uint8, device="cuda")
        return b

    def step(self):
        """One step of every clip; returns the segments it ran.  The resident clip is as long as the TIMED
        region (steps x frames-per-step frames); the warm-up steps run on its first frames and the timed
        steps carry on from there and wrap around, so every frame of the clip is encoded exactly once inside
        the timed region and the clip count does not depend on --warmup.  (S-coh / S-static clips do not wrap: they are
        as long as warm-up + timed region -- a wrap would put one complete redraw into the timed region; see main().)"""
        self.last_ops, segs = self.batch.encode_frames(self.fm, self.fa, self.args.frames_per_step, self.ops_buf, loop=True)
        return segs

    def first_ops(self, segs):
        """The opcodes of the step just run for the clips the oracle re-encodes afterwards -- the first, the middle and the
        last stream of the batch (Encoder.encode returns the rows as they were packed): {stream: (n, 6) tensor}."""
        return {s: self.last_ops[s].clone() for s in sorted({0, self.S // 2, self.S - 1})}

    def check(self):
        self.batch.enc.check()

    def profile(self, on):
        self.batch.enc.profile(on)

    def profile_read(self):
        return self.batch.enc.profile_read()

    def launch_forms(self):
        """greedy launches of the profiled (events) leg by kernel form (iiv_encoder_launch_forms)"""
        return self.batch.enc.launch_forms()

    def build_id(self):
        return self.native.build_id()

    def input_stats(self):
        """(share of the steps the nonces decided, form of the one-wave kernel the encoder has settled on) -- the encoder
        picks the form by what its kernels report about the input (include/iivision.h: iiv_encoder_input_stats)"""
        return self.batch.enc.input_stats()

    def uses_wave_kernel(self):
        return self.args.greedy != "workgroup" and not self.args.joint


class _QuietStdout:
    """The driver reads ONE JSON line from stdout; what libraries write there at C level (RCCL's version banner, gloo's
    connection notice) goes to stderr: file descriptor 1 points at stderr while the bench runs and is put back for the line."""

    def __init__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def restore(self):
        if self.saved is not None:
            sys.stdout.flush()
            os.dup2(self.saved, 1)
            os.close(self.saved)
            self.saved = None


def main(argv=None, backend_cls=None):
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and backend_cls is None:
        # no launcher around us: become one (before torch.distributed, libiivision.so or any HIP call)
        rc = launch_ranks(args, argv)
        if rc:
            raise SystemExit(rc)
        return None
    if backend_cls is None:
        backend_cls = _resolve_backend(args.backend)
    quiet = _QuietStdout()
    try:
        return _run(args, backend_cls, quiet)
    finally:
        quiet.restore()


def _run(args, backend_cls, quiet):
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (the launcher's rank count and --gpus must agree)" % (args.gpus, world))
    # N > 1: every rank on the CPUs of its GPU's NUMA node, before anything touches the GPU (IIV_BENCH_PIN=1 / 0 forces / forbids it)
    pin = None
    if os.environ.get("IIV_BENCH_PIN", "1" if world > 1 else "0") == "1":
        pin = pin_to_gpu_numa_node(0 if REHEARSE else local_rank)
    be = backend_cls(args, local_rank, world)
    # the process group exists for N > 1 only; IIV_BENCH_FORCE_DIST=1 creates it for one rank too, so that
    # the process-group initialisation (gloo, or RCCL with --dist-backend nccl), the barrier and the scalar reductions can be exercised on a 1-GPU box
    # (python -m torch.distributed.run --nproc-per-node 1 ... bench.py)
    use_dist = world > 1 or os.environ.get("IIV_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(be.dist_backend, **be.dist_kwargs())
    n_gpus = max(world, 1)
    cdev = getattr(be, "coll_device", be.device)   # where the three scalars that cross ranks live
    dhgr = args.mode == "DHGR"
    F = args.frames_per_step
    # resident clip length = the timed region: the warm-up runs on its first frames and the timed steps wrap around (every
    # frame encoded exactly once in the timed region).  Frame 0 after the last frame is an UNRELATED picture, which is what
    # S-iid and S-img frames are to each other anyway; for S-coh / S-static that wrap would put one complete redraw into the
    # timed region, so those clips are as long as warm-up + timed region and never wrap.
    wraps = not (args.coherent or args.static)
    n_frames = args.steps * F if wraps else (args.steps + args.warmup + events_leg_steps(args.steps)) * F
    S = args.streams
    if S <= 0:
        per_clip = n_frames * 8192 * (2 if dhgr else 1) + 300 * 1024 + F * OPS_PER_FRAME * 6   # frames + stream state (292 KB) + opcodes
        # 7168 one-wave streams are resident at a time (28 per CU): 14336 = two full rounds per launch
        # (+2 % over 12288, whose second round is 71 % full); it needs 246 of a free MI355X's 287 GiB
        S = next((c for c in (14336, 12288, 6144, 3072, 1536) if c * per_clip + (3 << 30) <= 0.88 * be.free_bytes()), 1536)
        if use_dist:   # every rank runs the same number of clips
            import torch
            t = torch.tensor([S], dtype=torch.int64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            S = int(t.item())

    # ---- setup (untimed): tables, synthetic clips, stream state, all in HBM
    t_tab = be.build_tables()
    be.make_clips(S, n_frames, data_seed(rank))
    seeds = rank_seeds(rank, S)
    be.make_batch(S, seeds)

    def barrier():
        if use_dist:
            dist.barrier()

    leg = timed_leg(be, args.steps, args.warmup, barrier)
    first_ops, op_count, seg_count, prof = leg["first_ops"], leg["op_count"], leg["seg_count"], leg["prof"]
    rank_elapsed = all_ranks(leg["elapsed"], cdev, world, use_dist)   # (one float per rank: what the line's per-rank rates come from)
    seed_lo = all_ranks(float(seeds[0][0]), cdev, world, use_dist)     # first / last stream seed of every rank: disjoint ranges
    seed_hi = all_ranks(float(seeds[-1][0]), cdev, world, use_dist)
    elapsed = max_over_ranks(leg["elapsed"], cdev, world, use_dist)
    # where every rank pinned itself (NUMA node, number of CPUs; -1 / 0: it ran unpinned)
    pin_node = all_ranks(float(pin["numa_node"] if pin and pin["pinned"] else -1), cdev, world, use_dist)
    pin_cpus = all_ranks(float(pin["n_cpus"] if pin and pin["pinned"] else 0), cdev, world, use_dist)

    frames_done = args.steps * F * S * n_gpus
    fps = frames_done / elapsed

    out = {
        "metric": "DHGR frames transcoded/sec" if dhgr else "HGR frames transcoded/sec",
        "value": fps,
        "unit": "frames/s",
        "n_gpus": n_gpus,
        "world_size": dist.get_world_size() if use_dist else 1,   # ranks the process group actually saw (1: no group)
        "dist_backend": (dist.get_backend() if use_dist else None),
        "launcher": os.environ.get("IIV_BENCH_LAUNCHED_BY") or ("torchrun" if "TORCHELASTIC_RUN_ID" in os.environ else "env" if world > 1 else None),
        "per_rank_frames_per_s": {"min": args.steps * F * S / max(rank_elapsed), "max": args.steps * F * S / min(rank_elapsed),
                                  "ranks": len(rank_elapsed)},
        "per_rank_ms_per_step": {"min": 1000.0 * min(rank_elapsed) / args.steps, "max": 1000.0 * max(rank_elapsed) / args.steps,
                                 "all": [round(1000.0 * e / args.steps, 3) for e in rank_elapsed]},
        "per_rank_stream_seeds": [[int(a), int(b)] for a, b in zip(seed_lo, seed_hi)],
        # NUMA node and CPU count every rank pinned itself to before its first GPU call (pin_to_gpu_numa_node; -1 / 0 = unpinned;
        # single-GPU runs do not pin unless IIV_BENCH_PIN=1); rank 0's own record in full
        "per_rank_cpu_affinity": {"numa_node": [int(x) for x in pin_node], "n_cpus": [int(x) for x in pin_cpus], "rank0": pin},
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1000.0 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u16",
        "data": "synthetic",
        "config": {
            "workload": "%s %s palette %dx192 S-%s synthetic clips, %d frames each, %d independent clips per GPU, "
                        "Movie.encode control flow (490 opcodes/frame%s)%s" % (
                            args.mode, "//gs RGB (IIGS)" if args.palette == "IIGS" else "NTSC", 560 if dhgr else 280,
                            "img" if args.img else ("static (2 %% redrawn per frame, x%d)" % args.repeat) if args.static else "coh" if args.coherent else "iid",
                            args.steps * F, S, ", bank flip per 2 KiB" if dhgr else "",
                            "; JOINT content choice (f4, not the reference's output)" if args.joint else
                            "; FOURTH offset per opcode (f4, not the reference's output)" if args.fourth else ""),
            "palette": args.palette,
            "streams_per_gpu": S,
            **({"greedy_form": be.input_stats()[1], "nonce_decided_share_of_steps": round(be.input_stats()[0], 4),
                "real_opcodes_per_stream_and_launch": round(getattr(getattr(getattr(be, "batch", None), "enc", None), "real_opcodes_per_launch", 0.0), 1)}
               if hasattr(be, "input_stats") and be.uses_wave_kernel() else {}),
            "frames_per_step": F,
            "resident_clip_frames": n_frames,
            "clip_wraps_in_timed_region": bool(wraps and args.warmup > 0),
            "opcodes_per_frame": OPS_PER_FRAME,
            "parallelism": ("REHEARSAL: %d ranks sharing ONE GPU (IIV_BENCH_REHEARSE_ON_ONE_GPU), %d streams each -- not a multi-GPU measurement"
                            if REHEARSE and world > 1 else "%d GPU x %d independent streams, no collective") % (n_gpus, S),
        },
        "opcodes_per_s": fps * OPS_PER_FRAME,
        # SURVEY 8(d)'s whole-pipeline figure: frames/s x algorithmic bytes per frame (prologue calls + opcodes) against the HBM peak
        "pipeline_roofline": {"bound": "hbm", "bytes_per_frame": BYTES_PER_FRAME[args.mode],
                              "achieved": fps / n_gpus * BYTES_PER_FRAME[args.mode] / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": fps / n_gpus * BYTES_PER_FRAME[args.mode] / 1e9 / HBM_PEAK_GBS, "per": "GPU"},
        "diff_weights": "table-gather" if args.dw_table or args.dw == "table" else args.dw,
        "table_build_s": t_tab,
        "build_id": be.build_id() if hasattr(be, "build_id") else None,   # iiv_version(): what committed counter runs must carry to be quoted
    }

    if rank == 0:
        ev = leg["events"]
        out.update(_roofline_objects(be, args, prof, ev["op_count"], ev["seg_count"], S, ev["elapsed"],
                                     live_ceiling=be.is_gpu and not args.no_extras))
        out["events_leg"] = {"steps": ev["steps"], "ms_per_step": 1000.0 * ev["elapsed"] / max(ev["steps"], 1),
                             "note": "untimed, between warm-up and timed region: the same steps with a HIP event pair around every kernel "
                                     "launch; the roofline objects' launch durations come from it, `value` from the timed region (no events)"}
        port_fps = None
        want = (lambda name: True) if args.extras == "all" else (lambda name, _w=set(args.extras.split(",")): name in _w)
        if n_gpus == 1 and not args.no_extras and be.is_gpu:
            if want("single_stream"):
                out["single_stream"] = _single_stream(be, args)
            if not args.no_emit and want("emit"):   # the same steps with the bytes leaving the device (PCIe-inclusive; never `value`)
                out["emit"] = _emit_end_to_end(be, args, fps)
            if want("make_data_tables"):
                out["make_data_tables_s"] = _make_data_tables_seconds()
        if not args.no_cpu_baseline and n_gpus == 1 and be.is_gpu:
            out["cpu_baseline"] = _cpu_baseline(be, seeds, args, ops_check=({s: t.cpu().numpy() for s, t in first_ops.items()}, F))
            out["cpu_baseline_all_cores"] = _cpu_baseline_all_cores(be, seeds, args)
            port_fps = out["cpu_baseline"]["value"]
            out["vs_reference_python"] = _vs_reference(args.mode, fps, port_fps)
            if "single_stream" in out:
                # what north_star's ">= 1000x at 1 GPU" reads against when ONE video is all there is
                # (BASELINE configs 3 / 4 as literally worded); the headline needs many clips per GPU
                r = _vs_reference(args.mode, out["single_stream"]["value"], port_fps)
                out["single_stream"]["vs_reference_python"] = r and r["value"]
                out["single_stream"]["note"] = ("one clip alone on one GPU; the >= 1000x target of north_star is met "
                                                "per GPU only with many independent clips (see value / vs_reference_python)")
        if n_gpus == 1 and not args.no_extras and be.is_gpu and not args.joint and not args.fourth and not args.no_emit and want("ingest"):
            try:
                out["ingest"] = _ingest_and_e2e(be, args, fps, out.get("emit", {}).get("value"))
            except Exception as e:   # (e.g. not enough free HBM beside other tenants: report, do not fail the line)
                out["ingest"] = {"value": None, "error": repr(e)}
        if n_gpus == 1 and not args.no_extras and be.is_gpu:
            if want("dropin"):
                out["dropin"] = _dropin_video(args)
            if dhgr and not args.joint and not args.fourth:   # SURVEY 8(d) M1 "plus HGR frames/s": a short HGR leg with its own tables and clips
                if want("hgr"):
                    out["hgr"] = _hgr_leg(be, args, local_rank, world)
                if not args.img and want("img"):   # picture-like input (S-img): what the product encodes, and the slowest input -- its own short leg
                    out["img"] = _hgr_leg(be, args, local_rank, world, steps=4, mode="DHGR", img=True)
                if want("fourth_offset"):
                    out["fourth_offset"] = _hgr_leg(be, args, local_rank, world, steps=4, mode="DHGR", fourth=True)
                if want("joint"):
                    out["joint"] = _hgr_leg(be, args, local_rank, world, steps=2, warmup=1, mode="DHGR", joint=True, streams=min(be.S, 3072))

        quiet.restore()
        print(json.dumps(out))
        sys.stdout.flush()
    if use_dist:
        dist.destroy_process_group()
    return out


def events_leg_steps(steps):
    """Steps of the untimed leg that runs with HIP events around every kernel launch (the per-kernel split of the roofline
    objects), between the warm-up and the timed region."""
    return min(int(steps), 3)


def timed_leg(be, steps, warmup, barrier=lambda: None):
    """W untimed warm-up steps, an untimed EVENTS leg, then exactly K steps bracketed by barrier + synchronize on both sides.
    The events leg (events_leg_steps(K) steps of the same work) is where the library records a HIP event pair around every
    kernel launch on the launch stream (iiv_encoder_profile): the kernels' average launch durations -- roofline.achieved,
    avg_launch_ms, kernel_time_share -- come from it, and the timed region that `value` is computed from runs with no
    event recording at all (round 5 recorded them inside it)."""
    first_ops = None   # stream 0's opcodes of the first step, checked against the oracle by _cpu_baseline
    for i in range(warmup):
        segs0 = be.step()
        if i == 0:
            first_ops = be.first_ops(segs0)
    be.check()
    # ---- the events leg
    be.profile(True)
    barrier()
    be.synchronize()
    e0 = time.perf_counter()
    ev_ops, ev_segs = 0, 0
    for i in range(events_leg_steps(steps)):
        segs = be.step()
        if first_ops is None and i == 0:   # (only when there is no warm-up step; async D2D copy)
            first_ops = be.first_ops(segs)
        ev_ops += sum(s[3] for s in segs)
        ev_segs += len(segs)
    be.synchronize()
    e1 = time.perf_counter()
    prof = be.profile_read()
    be.profile(False)
    be.check()
    # ---- the timed region
    barrier()
    be.synchronize()
    t0 = time.perf_counter()
    op_count, seg_count = 0, 0
    for i in range(steps):
        segs = be.step()
        op_count += sum(s[3] for s in segs)
        seg_count += len(segs)
    be.synchronize()
    barrier()
    t1 = time.perf_counter()
    be.check()
    return {"elapsed": t1 - t0, "prof": prof, "op_count": op_count, "seg_count": seg_count, "first_ops": first_ops,
            "events": {"steps": events_leg_steps(steps), "elapsed": e1 - e0, "op_count": ev_ops, "seg_count": ev_segs}}


def _gather_ceiling_live(S, mode="DHGR", shared=False):
    """The ceiling of the greedy step's access pattern, measured now: tools/gather_ceiling (built by
    __graft_entry__.build()) runs that pattern -- a streamed 1 KiB row + 8 divergent table loads per
    opcode, no arithmetic -- with as many waves as there are clips, in a child process.  HGR batches of 4096 clips
    and more run the LDS-shared form (sixteen streams per workgroup share the even bytes' L1 half in LDS): its
    pattern is the microbenchmark's variant E."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "gather_ceiling")
    try:
        env = dict(os.environ)
        if shared:
            env["IIV_GATHER_E"] = "1"
        r = subprocess.run([exe, "D", str(int(S))] + (["HGR"] if mode == "HGR" else []), capture_output=True, text=True, timeout=180, env=env)
        tag, waves, ms, gl = r.stdout.strip().split()[-4:]
        if r.returncode == 0 and tag == "D":
            if shared:
                e = [l for l in r.stdout.splitlines() if l.startswith("# E-HGR W=16")]
                ms_e = float(e[0].split(":")[1].split()[0])
                return float(S) * 490 * 512 / ms_e * 1e-6, ("tools/gather_ceiling variant E (HGR, W = 16: two of the eight loads from LDS) %s waves, run by "
                                                         "this bench.py beside the encode: %.4f ms per launch (variant D, every load from the L1: %s ms)" % (waves, ms_e, ms))
            return float(gl), "tools/gather_ceiling D %s%s, run by this bench.py beside the encode: %s ms per launch" % (
                waves, " HGR" if mode == "HGR" else "", ms)
    except Exception:
        pass
    return None, None


def _input_kind(args):
    """The synthetic input of a leg as the counter file names it: "iid", "img", or another kind (no counter run)."""
    if getattr(args, "static", False):
        return "static"
    if getattr(args, "img", False):
        return "img"
    if getattr(args, "coherent", False):
        return "coherent"
    return "iid"


PMC_LATEST = os.path.join(ROOT, "profiles", "pmc_latest.json")


def expected_greedy_kernel(mode, form, fourth=False):
    """The instantiation name tools/profile_summary.py records (`greedy_wave_kernel<mode, streams per workgroup, fourth>`) of
    the one-wave kernel form that ran most launches of this run: 1 = DHGR, 0 = HGR; plain form 1 stream per workgroup, the
    LDS-shared form 8 (DHGR) / 16 (HGR)."""
    m = 1 if mode == "DHGR" else 0
    w = 1 if form != "shared" else (8 if m else 16)
    return "greedy_wave_kernel<%d, %d, %s>" % (m, w, "true" if fourth else "false")


def _pmc_entry(mode, kind="iid", fourth=False, build_id=None, kernel=None):
    """The committed counter run (profiles/pmc_latest.json, tools/profile_summary.py) for this mode / input -- or why it must
    not be quoted: (entry or None, source text, status) with status "ok", "none" (no run of this mode / input / option is
    committed) or "stale" (the run was taken with another build of the library -- its build_id is not iiv_version()'s -- or
    its kernel is not the instantiation that ran most launches here).  The counters are NOT measured in this run; a kernel
    change that is not followed by a new counter run makes them stale, and a stale figure is reported as null."""
    key = mode if kind == "iid" else "%s:%s" % (mode, kind)
    try:
        with open(os.environ.get("IIV_PMC_LATEST", PMC_LATEST)) as f:
            allkeys = json.load(f)
    except Exception:
        return None, None, "none"
    if fourth or key not in allkeys:
        return None, "profiles/pmc_latest.json holds no counter run for %s%s (it has: %s)" % (
            key, " with the fourth offset" if fourth else "", ", ".join(sorted(allkeys))), "none"
    d = allkeys[key]
    if build_id is not None and d.get("build_id") != build_id:
        return None, "stale: profiles/pmc_latest.json's %s run was taken with build %s, this library is build %s (re-run tools/round_evidence.sh)" % (
            key, d.get("build_id", "<unstamped>"), build_id), "stale"
    if kernel is not None and d.get("kernel") != kernel:
        return None, "stale: profiles/pmc_latest.json's %s run profiled %s, this run's launches were %s" % (key, d.get("kernel"), kernel), "stale"
    src = "profiles/pmc_latest.json (%s at %d streams, build %s, bench args %s; per-stream bytes x this run's streams): a committed " \
          "counter run of this build, not this run" % (d.get("kernel", "?"), d.get("streams", 0), d.get("build_id", "?"), " ".join(d.get("bench_args", [])))
    return d, src, "ok"


def _roofline_objects(be, args, prof, op_count, seg_count, S, elapsed, live_ceiling):
    """roofline (greedy kernel, dominant), access_pattern_reference, roofline_prologue, kernel_time_share of one leg, from the
    HIP events the library records around its launches on the launch stream (timed_leg's events leg: op_count, seg_count
    and elapsed are that leg's).  achieved / frac are THIS run's measurement (SURVEY 8(d) algorithmic bytes over the launch
    time the events give); what a committed counter run adds sits under counter_* keys and `traffic`, is tagged with its
    build id, and is null -- traffic_source says "stale: ..." -- when that run is not of this build and kernel."""
    out = {}
    g_ms, g_n = prof["greedy_ms"], max(prof["greedy_launches"], 1)
    p_ms, p_n = prof["prologue_ms"], max(prof["prologue_launches"], 1)
    greedy_bytes = float(op_count) * S * BYTES_PER_OPCODE       # all launches of the leg
    achieved = greedy_bytes / (g_ms * 1e-3) / 1e9 if g_ms > 0 else 0.0
    fourth = getattr(args, "fourth", False)
    bid = be.build_id() if hasattr(be, "build_id") else None
    form = be.input_stats()[1] if hasattr(be, "input_stats") and be.uses_wave_kernel() else None
    forms = be.launch_forms() if hasattr(be, "launch_forms") else None
    if forms:   # the form that ran most launches of the leg
        form = max(("plain", "shared", "team", "workgroup"), key=lambda k: forms.get(k, 0))
    kernel = expected_greedy_kernel(args.mode, form, fourth) if (be.uses_wave_kernel() and form in ("plain", "shared")) else None
    entry, traffic_source, status = _pmc_entry(args.mode, _input_kind(args), fourth, bid, kernel) if be.uses_wave_kernel() else (None, None, "none")
    traffic = entry["greedy_hbm_bytes_per_launch_per_stream"] * S if entry else None
    out["roofline"] = {
        "kernel": "greedy_wave_kernel" if be.uses_wave_kernel() else "greedy_kernel",
        "kernel_instantiation": kernel,
        # What binds the kernel is NOT the quantity the HBM fraction below measures: a step is ~310 instructions per wave; at the
        # rate a register-only loop of the same vector : scalar mix issues at this residency (tools/issue_probe.hip,
        # profiles/r06_issue_probe.txt) issuing them takes `issue.issue_floor_frac` of the launch, the rest is dependent latency
        # the resident waves do not cover; HBM (`traffic_frac`) is under 0.4.  achieved / peak / frac stay the SURVEY 8(d) HBM
        # figure (algorithmic bytes over launch time), the contract's yardstick for the path.
        "bound": "issue",
        "frac_is": "algorithmic HBM bytes per launch / launch time / the 8 TB/s peak (SURVEY 8d) -- the contract's yardstick, not the binding resource",
        "issue": entry.get("issue") if entry else None,
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "measured_by": "HIP events around every launch on the launch stream, in an untimed leg of %s between warm-up and timed region" % (
            "the same steps" if not hasattr(be, "is_gpu") or be.is_gpu else "a stand-in"),
        "peak_measured_read": HBM_MEASURED_READ_GBS,
        "traffic": traffic,
        "traffic_source": traffic_source,
        "counters": status,
        "algorithmic_bytes_per_launch": greedy_bytes / g_n,
        "avg_launch_ms": g_ms / g_n,
        "launches": prof["greedy_launches"],
        "lookups_per_s": float(op_count) * S * 256 / (g_ms * 1e-3) if g_ms > 0 else 0.0,
        # HBM bytes the counters saw (committed run of this build) per launch / this run's launch time, against the peak: how busy HBM really is
        "traffic_frac": (traffic / (g_ms / g_n * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and g_ms > 0) else None,
        "launches_by_form": forms,
    }
    if getattr(args, "static", False):
        out["roofline"]["note"] = ("S-static: most opcodes are out-of-work padding (video.py:249-251), written 64 at a time without "
                                   "any scoring; they are counted at 534 B like real opcodes here, so achieved / frac overstate the kernel")
    if be.uses_wave_kernel():
        # A measured yardstick beside the datasheet one: tools/gather_ceiling.hip runs a step's ACCESS PATTERN -- a streamed
        # 1 KiB row + 8 divergent 2-byte table loads per opcode -- with no arithmetic at all.  It is a reference pattern, not a
        # ceiling (the kernel's LDS-shared form does some of those loads from LDS and can exceed it): no `peak`, no `frac`.
        loads = float(op_count) * S * 512 / (g_ms * 1e-3) / 1e9 if g_ms > 0 else 0.0
        ref, src = _gather_ceiling_live(S, args.mode, False) if live_ceiling else (None, None)
        if ref is None:
            ref, src = ((GATHER_CEILING_GLOADS_HGR, "tools/gather_ceiling D 14336 HGR, a run on an MI355X committed as a constant, not this run")
                        if args.mode == "HGR" else
                        (GATHER_CEILING_GLOADS, "profiles/r02l_gather_ceiling.txt, variant D at 12288 waves: a committed microbenchmark run, not this run"))
        out["access_pattern_reference"] = {
            "kernel": "greedy_wave_kernel", "kernel_form": form, "unit": "G table loads/s",
            "kernel_loads_per_s": loads, "reference_pattern_loads_per_s": ref, "kernel_over_reference": loads / ref,
            "reference_source": src,
            "note": "the reference is the step's bare access pattern with every load served by the L1 / TA (no arithmetic); it is a "
                    "measured yardstick, not an upper bound",
        }
    pro_bytes = float(seg_count) * S * BYTES_PER_PROLOGUE
    p_ach = pro_bytes / (p_ms * 1e-3) / 1e9 if p_ms > 0 else 0.0
    # the prologue of the same counter run; checked against the instantiation this run launched (mode, diff-weight form)
    p_kernel = "prologue_kernel<%d, %d>" % (1 if args.mode == "DHGR" else 0, {"table": 0, "recurrence": 1, "split": 2}.get(
        "table" if getattr(args, "dw_table", False) else getattr(args, "dw", "recurrence"), 1))
    p_entry, p_src, p_status = entry, traffic_source, status
    if p_entry is not None and p_entry.get("prologue_kernel") != p_kernel:
        p_entry, p_status = None, "stale"
        p_src = "stale: profiles/pmc_latest.json profiled %s, this run launched %s" % (entry.get("prologue_kernel"), p_kernel)
    if p_entry is None and p_status == "none" and not be.uses_wave_kernel():   # (the prologue is the same whatever the greedy kernel)
        p_entry, p_src, p_status = _pmc_entry(args.mode, _input_kind(args), fourth, bid, None)
        if p_entry is not None and p_entry.get("prologue_kernel") != p_kernel:
            p_entry, p_src, p_status = None, "stale: profiles/pmc_latest.json profiled %s, this run launched %s" % (p_entry.get("prologue_kernel"), p_kernel), "stale"
    p_traffic = p_entry["prologue_hbm_bytes_per_launch_per_stream"] * S if (p_entry and "prologue_hbm_bytes_per_launch_per_stream" in p_entry) else None
    c_ach = (p_traffic / (p_ms / p_n * 1e-3) / 1e9) if (p_traffic and p_ms > 0) else None
    out["roofline_prologue"] = {
        "kernel": "prologue_kernel",
        "kernel_instantiation": p_kernel,
        "bound": "hbm",
        # this run's measurement: SURVEY 8(d)'s 147 456 algorithmic bytes per call over the launch time the events give
        "achieved": p_ach,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": p_ach / HBM_PEAK_GBS,
        "frac_is": "SURVEY 8(d) algorithmic bytes per call / this run's launch time (HIP events) / the 8 TB/s peak",
        "avg_launch_ms": p_ms / p_n,
        "launches": prof["prologue_launches"],
        "algorithmic_bytes_per_launch": pro_bytes / p_n,
        # what the counters of a committed run OF THIS BUILD saw (the kernel moves fewer bytes than the SURVEY figure: 16-bit
        # priorities, recomputed diff weights), over this run's launch time; null when that run is stale
        "traffic": p_traffic,
        "traffic_source": p_src,
        "counters": p_status,
        "counter_achieved": c_ach,
        "counter_frac": (c_ach / HBM_PEAK_GBS) if c_ach else None,
        "peak_measured_copy": HBM_MEASURED_COPY_GBS,
        "counter_traffic_over_measured_copy": (c_ach / HBM_MEASURED_COPY_GBS) if c_ach else None,
    }
    out["kernel_time_share"] = {"greedy": g_ms / (1000 * elapsed), "prologue": p_ms / (1000 * elapsed),
                                "of": "the events leg (HIP events around every launch add host work the timed region does not have)"}
    return out


def _hgr_leg(be, args, local_rank, world, steps=6, warmup=1, mode="HGR", fourth=False, img=False, joint=False, streams=None):
    """HGR frames/s (BASELINE config 3's workload) in the default line: the DHGR leg's clips and tables are
    released, HGR tables are built and `steps` x 50 frames of as many HGR S-iid clips are encoded the same way.
    (mode="DHGR", fourth=True: the same short leg for f4's fourth offset per opcode -- not the reference's stream.)"""
    import copy
    import gc
    S = streams or be.S
    for name in ("fm", "fa", "batch", "ops_buf", "last_ops", "table", "store"):
        if name == "batch" and getattr(be, "batch", None) is not None:
            be.batch.close()
        setattr(be, name, None)
    gc.collect()
    be.torch.cuda.empty_cache()
    a2 = copy.copy(args)
    a2.mode, a2.steps, a2.warmup, a2.fourth, a2.joint = mode, steps, warmup, fourth, joint
    a2.img, a2.coherent, a2.static, a2.img_distinct = img, False, False, 2048
    h = GpuBackend(a2, local_rank, world)
    h.build_tables()
    n_frames = steps * a2.frames_per_step
    h.make_clips(S, n_frames, data_seed(0) + 1)
    h.make_batch(S, rank_seeds(0, S))
    leg = timed_leg(h, steps, warmup)
    fps = steps * a2.frames_per_step * S / leg["elapsed"]
    out = {"metric": "%s frames transcoded/sec" % mode, "value": fps, "unit": "frames/s", "steps": steps, "warmup": warmup,
           "ms_per_step": 1000.0 * leg["elapsed"] / steps,
           "workload": "%s NTSC palette S-%s synthetic clips, %d independent clips x %d frames, Movie.encode "
                       "control flow (490 opcodes/frame)%s" % (mode, "img" if img else "iid", S, n_frames,
                                                                " (2048 distinct picture-like clips tiled over the streams, every "
                                                                "stream with its own RNG seeds)" if img and S > 2048 else "")}
    if h.uses_wave_kernel():
        share, form = h.input_stats()
        out["nonce_decided_share_of_steps"] = round(share, 4)
        out["greedy_form"] = form
    if joint:
        out["note"] = ("IIV_CONTENT_JOINT (SURVEY 8 f4, README.md:212-215): every step's content byte chosen jointly with its extra "
                       "offsets -- 128 x the lookups of a reference step, two byte values per packed 16-bit instruction in the "
                       "256-thread workgroup kernel (DESIGN.md 7b); NOT the reference's stream, off by default")
    elif fourth:
        o = leg["first_ops"][0].cpu().numpy().reshape(-1, 6)[:, 2:6]
        o.sort(axis=1)
        out["distinct_offsets_per_opcode"] = float(((o[:, 1:] != o[:, :-1]).sum(axis=1) + 1).mean())
        out["note"] = ("IIV_OPT_FOURTH_OFFSET: up to three extra offsets per opcode instead of the reference's two and a copy of "
                       "the first (video.py:146,180-186) -- NOT the reference's stream, off by default; what it buys in picture "
                       "error per frame: profiles/r03_fourth_offset.txt")
    else:
        ev = leg["events"]
        out.update(_roofline_objects(h, a2, leg["prof"], ev["op_count"], ev["seg_count"], S, ev["elapsed"], live_ceiling=True))
    h.batch.close()
    return out


def _dropin_video(args, n_frames=20):
    """The drop-in path: the reference's own calling convention -- video.Video(...).encode_frame(target, is_aux)
    pulled one opcode per next() from Python, as movie.Movie.encode does (movie.py:56-111) -- for one clip.
    `value` = as it comes (Video.SPECULATE batches launches behind the generator); `with_budget` = the caller
    passes encode_frame(..., budget=K), the one optional keyword the mirror adds."""
    import contextlib
    import io
    import random
    import numpy as np
    import palette
    import screen
    import stream_batch
    import video
    import video_mode
    dhgr = args.mode == "DHGR"
    pal = palette.Palette(PALETTE_IDS[args.palette])
    fm, fa = stream_batch.synth_frames_torch(1, n_frames, dhgr, seed=3, device="cpu")

    class FrameGrabber:
        input_frame_rate = 30

    def target_of(tgts, fr):
        if fr not in tgts:
            main = screen.MemoryMap(1, fm[0, fr].numpy().copy())
            tgts[fr] = (screen.DHGRBitmap(main_memory=main, aux_memory=screen.MemoryMap(1, fa[0, fr].numpy().copy()), palette=pal)
                        if dhgr else screen.HGRBitmap(main_memory=main, palette=pal))
        return tgts[fr]

    def run(budget, lookahead=True, live=True):
        random.seed(1)
        np.random.seed(1)
        v = video.Video(FrameGrabber(), ticks_per_second=14700., palette=pal,
                        mode=video_mode.VideoMode.DHGR if dhgr else video_mode.VideoMode.HGR)
        v.LOOKAHEAD = lookahead
        v.LIVE = live
        tgts = {}
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            if budget:
                for (fr, ia, _, k) in stream_batch.MovieClock(dhgr).segments(n_frames):
                    gen = v.encode_frame(target_of(tgts, fr), is_aux=bool(ia), budget=k)
                    for _ in range(k):
                        next(gen)
            else:
                # movie.Movie.encode + emit_stream, statement by statement (movie.py:56-150), audio and opcode objects left out
                ticks, stream_pos, aux, last_bank, op_seq, target = 0, 7, False, False, None, None
                while True:
                    ticks += 1
                    if v.tick(ticks):
                        if v.frame_number - 1 >= n_frames:
                            break
                        target = target_of(tgts, v.frame_number - 1)
                        op_seq = v.encode_frame(target, is_aux=aux)
                        v.out_of_work = {True: False, False: False}
                    if aux != last_bank:
                        last_bank = aux
                        op_seq = v.encode_frame(target, is_aux=aux)
                    next(op_seq)
                    stream_pos += 7
                    if stream_pos % 2048 >= 2044:
                        if dhgr:
                            aux = not aux
                        stream_pos += 4
        dt = time.perf_counter() - t0
        stats.update(getattr(v, "lookahead_stats", {}))
        ls = getattr(v, "live_stats", None)
        if ls and live and not budget:
            live_stats.update({"launches": ls["launches"], "polls": ls["takes"], "polls_that_waited": ls["waits"],
                               "us_per_frame_waiting_for_the_device": round(1e6 * ls["wait_s"] / n_frames, 1)})
        return n_frames / dt

    stats, live_stats = {}, {}
    try:
        run(False)   # (warm-up: table build, first launches)
        return {"value": run(False), "lookahead": dict(stats), "live": dict(live_stats),
                "without_live": run(False, live=False), "without_lookahead": run(False, lookahead=False), "with_budget": run(True),
                "unit": "frames/s", "frames": n_frames,
                "what": "video.Video driven from Python as movie.Movie.encode drives it (tick() per audio sample, a generator per "
                        "frame and bank flip, one next() per opcode), %s, one clip; lookahead: the generators behind a bank flip enqueued ahead of the caller "
                        "(Video.LOOKAHEAD) and what became of them; live: the opcodes handed out while the kernel produces them (Video.LIVE, "
                        "iiv_encode_live: a queue in coherent host memory) -- launches, polls of the queue and how long they waited; "
                        "with_budget: encode_frame(..., budget=k)" % args.mode}
    except Exception as e:
        return {"value": None, "error": repr(e)}


def _vs_reference(mode, gpu_fps, port_fps_here):
    """Speed-up over the reference's own Python: the reference cannot run on the GPU box, so
    (GPU / port on this box) x (port / reference, both measured in the build container by
    tools/measure_reference_ratio.py and committed as profiles/reference_ratio.json)."""
    try:
        with open(os.path.join(ROOT, "profiles", "reference_ratio.json")) as f:
            d = json.load(f)
        r = d[mode]
    except Exception:
        return None
    return {
        "value": gpu_fps / port_fps_here * r["port_over_reference"],
        "gpu_over_port_on_this_box": gpu_fps / port_fps_here,
        "port_over_reference_python": r["port_over_reference"],
        "reference_python_frames_per_s_in_build_container": r["reference_python_frames_per_s"],
        "provenance": "profiles/reference_ratio.json: tools/measure_reference_ratio.py, %s, %s, measured %s" % (
            d.get("model", "?"), d.get("workload", "?"), d.get("measured", "?")),
    }


def _single_stream(be, args):
    """One clip alone (a video is a sequential chain: latency-bound).  A 50-frame step of one clip is 16 ms of GPU time
    behind 268 launches: one host hiccup moves it by 10 %, so three such steps are timed and the median is reported."""
    import torch

    def rate(fm, fa):
        b = be.sb.StreamBatch(be.mode, be.table, be.store, 1, seeds=[(1, 1)], dm=be.dm, joint_content=args.joint, fourth_offset=args.fourth)
        b.enc.set_greedy_kernel({"auto": None, "wave": True, "workgroup": False}.get(args.greedy, args.greedy))
        b.encode_frames(fm, fa, 10)
        torch.cuda.synchronize()
        dts = []
        for _ in range(3):
            t0 = time.perf_counter()
            b.encode_frames(fm, fa, 50)
            torch.cuda.synchronize()
            dts.append(time.perf_counter() - t0)
        b.enc.check()
        b.close()
        return sorted(dts)[1]

    fm, fa = be.sb.synth_frames_torch(1, 160, be.dhgr, seed=99, coherent=args.coherent)
    dt = rate(fm, fa)
    out = {"value": 50 / dt, "unit": "frames/s", "us_per_opcode": 1e6 * dt / (50 * OPS_PER_FRAME),
           "sample": "median of three consecutive 50-frame steps of one 160-frame clip (after 10 warm-up frames)"}
    # the same on picture-like input (S-img): there the nonces decide nearly every step's extra offsets (96 % of the
    # opcodes), which used to end the eight-wave kernel's run of concurrent steps at each of them
    try:
        fm, fa = be.sb.synth_frames_img(1, 160, be.dhgr, seed=99)
        out["value_img"] = 50 / rate(fm, fa)
    except Exception as e:
        out["value_img"] = None
        out["value_img_error"] = repr(e)
    return out


def _make_data_tables_seconds():
    """BASELINE metric M2: make_data_tables.main() -- both palettes, HGR + DHGR, four .npz files in the
    reference's own format (lower triangle, key edit_distance) -- in a scratch directory."""
    import shutil
    import tempfile
    import make_data_tables
    cwd = os.getcwd()
    d = tempfile.mkdtemp(prefix="iiv_tables_")
    try:
        os.chdir(d)
        import contextlib
        import io
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            make_data_tables.main()
        dt = time.perf_counter() - t0
        files = sorted(os.listdir(os.path.join(d, make_data_tables.DATA_DIR)))
        assert len(files) == 4, files
        return {"value": dt, "unit": "s", "files": files,
                "bytes": sum(os.path.getsize(os.path.join(d, make_data_tables.DATA_DIR, f)) for f in files),
                "reference": "README.md:67 'about 90 minutes on my machine'"}
    except Exception as e:   # (e.g. a scratch disk too small for 3 GiB: report, do not fail the line)
        return {"value": None, "error": repr(e)}
    finally:
        os.chdir(cwd)
        shutil.rmtree(d, ignore_errors=True)


def _emit_end_to_end(be, args, resident_fps):
    """The same workload with the output leaving the device: encode (iiv_encode) -> .a2m bytes
    (iiv_emit_chunk, header / ACK framing, silence as the audio) -> pinned host memory, the copy of
    step k overlapping the encode of step k + 1.  A fresh batch over the same clips."""
    import numpy as np
    import torch
    native = be.native
    S, F = be.S, args.frames_per_step
    rng = np.random.default_rng(0)
    tick_addr = torch.from_numpy(rng.integers(0x4000, 0x7fff, 1024).astype(np.int16)).cuda()   # (addresses are data)
    b = be.sb.StreamBatch(be.mode, be.table, be.store, S, seeds=rank_seeds(0, S), dm=be.dm)
    n_ops = F * OPS_PER_FRAME
    ops = torch.empty((S, n_ops, 6), dtype=torch.uint8, device="cuda")
    width = native.emit_chunk_range(be.mode, 0, n_ops)[1] + 16
    dev = [torch.empty(S * width, dtype=torch.uint8, device="cuda") for _ in range(2)]
    host = [torch.empty(S * width, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
    _dummies = [torch.cuda.Stream() for _ in range(int(os.environ.get("IIV_BENCH_DUMMY_STREAMS", "0")))]   # (experiment: HIP stream -> hardware queue mapping)
    copy_stream = torch.cuda.Stream()
    done = [torch.cuda.Event(), torch.cuda.Event()]
    ready = [torch.cuda.Event(), torch.cuda.Event()]
    state = {"first_op": 0, "bytes": 0}

    def step(k):
        view, segs = b.encode_frames(be.fm, be.fa, F, ops, loop=True)   # (S, n, 6): the rows as they were packed
        n = sum(s[3] for s in segs)
        j = k & 1
        torch.cuda.current_stream().wait_event(done[j])          # the copy that last read dev[j] has finished
        nb = native.emit_chunk_range(be.mode, state["first_op"], n)[1]
        out = dev[j][: S * nb].view(S, nb)                        # rows exactly as long as the slice: one contiguous copy
        native.emit_chunk(be.mode, view, state["first_op"], tick_addr, 0xBA72, out)
        ready[j].record()
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(ready[j])
            host[j][: S * nb].copy_(dev[j][: S * nb], non_blocking=True)
            done[j].record()
        state["first_op"] += n
        state["bytes"] += nb * S

    for ev in done:
        ev.record()
    for k in range(args.warmup):
        step(k)
    torch.cuda.synchronize()
    state["bytes"] = 0
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    b.enc.check()
    b.close()
    fps = args.steps * F * S / dt
    return {"value": fps, "unit": "frames/s", "vs_resident_only": fps / resident_fps,
            "bytes_to_host": state["bytes"], "d2h_gb_per_s": state["bytes"] / dt / 1e9,
            "what": "iiv_encode -> iiv_emit_chunk (.a2m framing, tick 34) -> hipMemcpyAsync to pinned host memory, "
                    "double-buffered; %d steps" % args.steps}


INGEST_BYTES_PER_FRAME = {"DHGR": 192 * 280 * 3 + 2 * 8192, "HGR": 192 * 280 * 3 + 8192}   # 161 KB of RGB in, 8 / 16 KiB of memory maps out


def _ingest_and_e2e(be, args, resident_fps, emit_fps):
    """SURVEY 8(f3) with numbers, and the pipeline from pixels to bytes:
      ingest   iiv_frames_to_memory_maps alone, as many frames per step as the batch encodes (S x F), ordered dither and error
               diffusion: frames/s and the fraction of the HBM peak its 161 KB in + 8 / 16 KiB out per frame amount to;
      e2e      RGB frames -> iiv_frames_to_memory_maps -> iiv_encode -> iiv_emit_chunk -> pinned host memory, every step on
               fresh frames, the conversion of step k + 1 on a second HIP stream beside the encode of step k.
    The RGB source is `distinct` synthetic picture-like clips (stream_batch.synth_rgb_torch) tiled over the S streams -- every
    tile is converted again (S x F conversions per step, nothing cached), every stream keeps its own RNG seeds.  The DHGR
    leg's clips are released first: the source (K x distinct x F frames of 161 KB) and two (S, F, 32, 256) target buffers per
    bank take their place."""
    import gc
    import numpy as np
    import torch
    import palette
    native = be.native
    S, F = be.S, args.frames_per_step
    for name in ("fm", "fa", "ops_buf", "last_ops"):
        setattr(be, name, None)
    if getattr(be, "batch", None) is not None:
        be.batch.close()
        be.batch = None
    gc.collect()
    torch.cuda.empty_cache()
    K = 4                                  # steps of distinct source frames
    distinct = min(512, S)
    pal_rgb = palette.PALETTES[palette.Palette(PALETTE_IDS[args.palette])].rgb_array()
    rgb = be.sb.synth_rgb_torch(distinct, K * F, seed=data_seed(0) + 5)            # (distinct, K F, 192, 280, 3)
    rgb = rgb.view(distinct, K, F, 192, 280, 3).transpose(0, 1).contiguous()        # (K, distinct, F, ...): a step's source is contiguous
    torch.cuda.empty_cache()
    nb = 2 if be.dhgr else 1
    bufs = [[torch.empty((S, F, 32, 256), dtype=torch.uint8, device="cuda") for _ in range(nb)] for _ in range(2)]

    def convert(k, j, dither):
        """step k's frames of every stream into buffer j: one call per tile of `distinct` streams"""
        src = rgb[k % K].view(distinct * F, 192, 280, 3)
        for s0 in range(0, S, distinct):
            n = min(distinct, S - s0)
            native.frames_to_memory_maps(be.mode, pal_rgb, src[: n * F], dither,
                                         out=(bufs[j][0][s0:s0 + n], bufs[j][1][s0:s0 + n] if be.dhgr else None))

    out = {"frames_per_step": S * F, "distinct_source_clips": distinct,
           "bytes_per_frame": INGEST_BYTES_PER_FRAME[args.mode],
           "what": "iiv_frames_to_memory_maps (replaces the external bmp2dhr, frame_grabber.py:68-115; no reference output exists), "
                   "%d x %d frames per step, %d distinct synthetic RGB clips tiled over the streams" % (S, F, distinct)}
    for name, dither in (("ordered", 32), ("diffusion", native.DITHER_DIFFUSION)):
        convert(0, 0, dither)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for k in range(K):
            convert(k, k & 1, dither)
        ev1.record()
        torch.cuda.synchronize()
        dt = ev0.elapsed_time(ev1) * 1e-3
        fps = K * S * F / dt
        gbs = fps * INGEST_BYTES_PER_FRAME[args.mode] / 1e9
        out[name] = {"value": fps, "unit": "frames/s", "ms_per_step": 1e3 * dt / K,
                     "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                  "peak_measured_copy": HBM_MEASURED_COPY_GBS, "over_measured_copy": gbs / HBM_MEASURED_COPY_GBS},
                     "vs_encoder_rate": fps / resident_fps}

    # The copy and the conversion run on two side streams, made ONCE for all the e2e legs: HIP spreads streams over a handful
    # of hardware queues in the order they are made, and a leg that made its own pair could find its copy stream on the
    # encode's own queue -- the 45 ms copy of a step's 2.5 GB then ran between two launches instead of beside them, in one
    # leg or another from run to run (round 6: emit_same_content read 2.80 or 3.30 M frames/s, its diffusion twin the other).
    side_streams = (torch.cuda.Stream(), torch.cuda.Stream())

    def e2e(dither, overlap, pre=None):
        """pre: the steps' frames already converted (a list of buffers like bufs[j]) -- the same pipeline WITHOUT the conversion,
        on the same content: what e2e's rate is to be compared with"""
        rng = np.random.default_rng(0)
        tick_addr = torch.from_numpy(rng.integers(0x4000, 0x7fff, 1024).astype(np.int16)).cuda()
        b = be.sb.StreamBatch(be.mode, be.table, be.store, S, seeds=rank_seeds(0, S), dm=be.dm)
        n_ops = F * OPS_PER_FRAME
        ops = torch.empty((S, n_ops, 6), dtype=torch.uint8, device="cuda")
        width = native.emit_chunk_range(be.mode, 0, n_ops)[1] + 16
        dev = [torch.empty(S * width, dtype=torch.uint8, device="cuda") for _ in range(2)]
        host = [torch.empty(S * width, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
        copy_stream, ingest_stream = side_streams      # (made once for all the legs below: see there)
        done = [torch.cuda.Event(), torch.cuda.Event()]
        ready = [torch.cuda.Event(), torch.cuda.Event()]
        converted = [torch.cuda.Event(), torch.cuda.Event()]     # buffer j holds its step's frames
        encoded = [torch.cuda.Event(), torch.cuda.Event()]       # the encode that read buffer j has finished
        state = {"first_op": 0, "bytes": 0}
        main = torch.cuda.current_stream()

        def ingest(k):
            j = k & 1
            if pre is not None:
                return
            if overlap:
                with torch.cuda.stream(ingest_stream):
                    ingest_stream.wait_event(encoded[j])
                    convert(k, j, dither)
                    converted[j].record()
            else:
                convert(k, j, dither)

        def step(k):
            j = k & 1
            src = bufs[j] if pre is None else pre[k % len(pre)]
            if pre is not None:
                pass
            elif overlap:
                main.wait_event(converted[j])
                if k + 1 <= K:
                    ingest(k + 1)                                 # the next step's frames, beside this step's encode
            else:
                ingest(k)
            view, segs = b.encode_frames(src[0], src[1] if be.dhgr else None, F, ops, loop=True)
            encoded[j].record()
            n = sum(s_[3] for s_ in segs)
            main.wait_event(done[j])
            nbytes = native.emit_chunk_range(be.mode, state["first_op"], n)[1]
            o = dev[j][: S * nbytes].view(S, nbytes)
            native.emit_chunk(be.mode, view, state["first_op"], tick_addr, 0xBA72, o)
            ready[j].record()
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(ready[j])
                host[j][: S * nbytes].copy_(dev[j][: S * nbytes], non_blocking=True)
                done[j].record()
            state["first_op"] += n
            state["bytes"] += nbytes * S

        for ev in done + encoded:
            ev.record()
        if overlap and pre is None:
            ingest(0)
        step(0)                                                   # warm-up step (its frames are step 0's)
        torch.cuda.synchronize()
        state["bytes"] = 0
        t0 = time.perf_counter()
        for k in range(1, 1 + K):
            step(k)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        b.enc.check()
        share, form = b.enc.input_stats()
        b.close()
        fps = K * F * S / dt
        return {"value": fps, "unit": "frames/s", "ms_per_step": 1e3 * dt / K, "steps": K,
                "vs_emit": (fps / emit_fps) if emit_fps else None, "vs_resident_only": fps / resident_fps,
                "d2h_gb_per_s": state["bytes"] / dt / 1e9, "greedy_form": form, "nonce_decided_share_of_steps": round(share, 4)}

    out["e2e"] = e2e(32, True)
    out["e2e"]["what"] = ("RGB -> iiv_frames_to_memory_maps (ordered dither, amplitude 32) -> iiv_encode -> iiv_emit_chunk -> pinned host "
                          "memory; the conversion of step k + 1 runs on a second HIP stream beside the encode of step k; "
                          "vs_emit compares with the `emit` leg's rate on S-iid memory maps (different content: these frames are "
                          "picture-like and encode more slowly), vs_emit_same_content with the same pipeline on THESE frames "
                          "converted beforehand (emit_same_content)")
    out["e2e_diffusion"] = e2e(native.DITHER_DIFFUSION, True)
    out["e2e_serial"] = e2e(32, False)
    out["e2e_serial"]["what"] = "the same with the conversion on the encode's own stream (no overlap)"
    # the pipeline behind the frames alone, on the same content: the K steps' frames converted beforehand
    try:
        pre = []
        for k in range(K):
            pre.append([torch.empty((S, F, 32, 256), dtype=torch.uint8, device="cuda") for _ in range(nb)])
            keep = bufs[0]
            bufs[0] = pre[k]
            convert(k, 0, 32)
            bufs[0] = keep
        torch.cuda.synchronize()
        out["emit_same_content"] = e2e(32, False, pre=pre)
        out["emit_same_content"]["what"] = "iiv_encode -> iiv_emit_chunk -> pinned host memory on the e2e leg's own frames, converted beforehand"
        out["e2e"]["vs_emit_same_content"] = out["e2e"]["value"] / out["emit_same_content"]["value"]
        out["e2e_serial"]["vs_emit_same_content"] = out["e2e_serial"]["value"] / out["emit_same_content"]["value"]
        # ... and the same for the frames error diffusion makes of the source (other content: other rate)
        for k in range(K):
            keep = bufs[0]
            bufs[0] = pre[k]
            convert(k, 0, native.DITHER_DIFFUSION)
            bufs[0] = keep
        torch.cuda.synchronize()
        out["emit_same_content_diffusion"] = e2e(native.DITHER_DIFFUSION, False, pre=pre)
        out["emit_same_content_diffusion"]["what"] = "the same on the error-diffusion frames of e2e_diffusion, converted beforehand"
        out["e2e_diffusion"]["vs_emit_same_content"] = out["e2e_diffusion"]["value"] / out["emit_same_content_diffusion"]["value"]
        del pre
    except Exception as e:   # (e.g. not enough free HBM for the K pre-converted steps)
        out["emit_same_content"] = {"value": None, "error": repr(e)}
    del rgb, bufs
    gc.collect()
    torch.cuda.empty_cache()
    return out


def _cpu_baseline(be, seeds, args, ops_check):
    """Oracle (single-thread C port of video.py/screen.py) on stream 0's first frames -- the timed sample -- and, as the
    checker, on the first step's frames of every stream whose GPU opcodes `ops_check` holds (first / middle / last stream
    of the batch: the 14336-stream launches of the persistent workgroups, sampled across the queue)."""
    import numpy as np
    import oracle as O
    import stream_batch
    fm, fa = be.fm, be.fa
    n = min(args.cpu_frames, fm.shape[1])
    _, dm = O.cie2000_matrix(O.PALETTE_RGB[PALETTE_IDS[args.palette]])
    tab = O.build_table(be.mode, dm, symmetric=True)   # untimed, like the GPU's table build

    def encode(stream, n_fr):
        main = fm[stream, :n_fr].cpu().numpy()
        aux = fa[stream, :n_fr].cpu().numpy() if fa is not None else None
        v = O.Video(be.mode, tab, seed_py=seeds[stream][0], seed_np=seeds[stream][1])
        v.set_joint(args.joint)
        v.set_fourth_offset(args.fourth)
        segs = stream_batch.MovieClock(be.dhgr).segments(n_fr)
        t0 = time.perf_counter()
        got = []
        for (fr, ia, restart, k) in segs:
            if restart:
                v.encode_frame(main[fr], aux[fr] if aux is not None else None, ia)
            got.append(v.next(k))
        return np.concatenate(got), time.perf_counter() - t0

    cpu0, dt = encode(0, n)
    parity = None
    if ops_check is not None:
        # the oracle as the checker: the GPU's opcode streams of these clips' first frames, bit for bit
        gpu, nf = ops_check
        nf = int(min(nf, n))
        per = []
        for stream in sorted(gpu):
            g = gpu[stream]
            c = cpu0 if stream == 0 else encode(stream, nf)[0]
            c = c[: g.shape[0]]
            per.append({"stream": int(stream), "opcodes": int(c.shape[0]),
                        "equal": bool(c.shape == g[: c.shape[0]].shape and (c == g[: c.shape[0]]).all())})
        parity = {"streams": [p_["stream"] for p_ in per], "frames": nf, "opcodes": int(sum(p_["opcodes"] for p_ in per)),
                  "equal": all(p_["equal"] for p_ in per), "per_stream": per}
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


def _cpu_baseline_all_cores(be, seeds, args):
    """One independent stream per host thread (the oracle's C calls release the GIL)."""
    import concurrent.futures
    import oracle as O
    import stream_batch
    fm, fa = be.fm, be.fa
    threads = min(os.cpu_count() or 1, fm.shape[0])
    n = min(args.cpu_frames_all, fm.shape[1])
    main = fm[:threads, :n].cpu().numpy()
    aux = fa[:threads, :n].cpu().numpy() if fa is not None else None
    _, dm = O.cie2000_matrix(O.PALETTE_RGB[PALETTE_IDS[args.palette]])
    tab = O.build_table(be.mode, dm, symmetric=True)
    segs = stream_batch.MovieClock(be.dhgr).segments(n)
    vids = [O.Video(be.mode, tab, seed_py=seeds[i][0], seed_np=seeds[i][1]) for i in range(threads)]
    for v in vids:
        v.set_joint(args.joint)
        v.set_fourth_offset(args.fourth)

    def work(i):
        v = vids[i]
        for (fr, ia, restart, k) in segs:
            if restart:
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
