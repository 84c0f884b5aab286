"""CPU, world_size 2 over gloo: the N>1 path of bench.py shards streams across ranks with no
data-path collective; the only cross-rank traffic is the barrier, the MIN-reduce of the clip
count and the MAX-reduce of the elapsed time.

bench.main() itself is executed in both ranks -- rank / world logic, clip-count agreement,
seeding, warm-up, timed loop, reductions, the JSON line of rank 0, teardown -- with the device
work replaced at the bench.GpuBackend boundary by a CPU stand-in (there is no GPU here and no
CPU fallback of the product path; the stand-in only counts what it is asked to do)."""

import json
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"] = str(rank)
    os.environ["WORLD_SIZE"] = str(world)
    sys.path.insert(0, ROOT)
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S = 4
    seeds = bench.rank_seeds(rank, S)
    elapsed = bench.max_over_ranks(1.0 + rank, torch.device("cpu"), world)
    dist.barrier()
    q.put((rank, seeds, elapsed, bench.data_seed(rank)))
    dist.destroy_process_group()


def _spawn(target, world, extra=()):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29511 + os.getpid() % 200
    procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(extra)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_two_ranks_shard_streams_and_reduce_time():
    (r0, s0, e0, d0), (r1, s1, e1, d1) = _spawn(_worker, 2)
    assert e0 == e1 == 2.0                       # MAX over ranks
    assert not set(s0) & set(s1)                  # disjoint RNG seeds => independent streams
    assert d0 != d1                               # different synthetic clips per rank


class _CpuStandIn:
    """bench.GpuBackend's interface without a device: records the calls of bench.main()."""

    dist_backend = "gloo"
    is_gpu = False

    def __init__(self, args, local_rank, world):
        import stream_batch
        self.args = args
        self.device = torch.device("cpu")
        self.dhgr = args.mode == "DHGR"
        self.clock = stream_batch.MovieClock(self.dhgr)
        self.rank = int(os.environ["RANK"])
        self.log = {"steps": 0, "checks": 0, "sync": 0}

    def dist_kwargs(self):
        return {}

    def free_bytes(self):
        # rank 1 pretends to have less free memory: the ranks must agree on the smaller clip count
        return (240 << 30) if self.rank == 0 else (5 << 30)

    def synchronize(self):
        self.log["sync"] += 1

    def build_tables(self):
        return 0.0

    def make_clips(self, S, n_frames, seed):
        self.log["clips"] = (S, n_frames, seed)

    def make_batch(self, S, seeds):
        self.S = S
        self.log["seeds"] = (seeds[0], seeds[-1], len(seeds))

    def step(self):
        self.log["steps"] += 1
        return self.clock.segments(self.args.frames_per_step)

    def first_ops(self, segs):
        return torch.zeros((sum(s[3] for s in segs), 6), dtype=torch.uint8)

    def check(self):
        self.log["checks"] += 1

    def profile(self, on):
        pass

    def profile_read(self):
        return {"prologue_ms": 1.0, "greedy_ms": 2.0, "prologue_launches": 1, "greedy_launches": 1}

    def uses_wave_kernel(self):
        return True


def _bench_worker(rank, world, port, q, argv):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"] = str(rank)
    os.environ["LOCAL_RANK"] = str(rank)
    os.environ["WORLD_SIZE"] = str(world)
    for p in (ROOT, os.path.join(ROOT, "ii-vision_amd", "transcoder")):
        sys.path.insert(0, p)
    import contextlib
    import io
    import bench
    made = []

    def factory(args, local_rank, world_):
        made.append(_CpuStandIn(args, local_rank, world_))
        return made[-1]

    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = bench.main(argv, backend_cls=factory)
    q.put((rank, out["value"], out["config"], made[0].log, buf.getvalue()))


def test_bench_main_runs_in_two_ranks():
    argv = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--frames-per-step", "4", "--palette", "IIGS"]
    (r0, v0, cfg0, log0, line0), (r1, v1, cfg1, log1, line1) = _spawn(_bench_worker, 2, (argv,))
    # both ranks settled on the clip count the smaller rank can hold (MIN over ranks), and the
    # whole-job value counts both GPUs
    assert cfg0["streams_per_gpu"] == cfg1["streams_per_gpu"] == log0["clips"][0] == log1["clips"][0]
    assert cfg0["streams_per_gpu"] < 12288
    assert v0 == v1 and v0 > 0
    assert cfg0["palette"] == "IIGS" and "IIGS" in cfg0["workload"] and "2 GPU" in cfg0["parallelism"]
    # disjoint seeds and clips per rank; warm-up + timed steps ran; state was checked on both sides
    assert log0["seeds"][0] != log1["seeds"][0] and log0["clips"][2] != log1["clips"][2]
    assert log0["steps"] == log1["steps"] == 4 and log0["checks"] == 2 and log0["sync"] >= 2
    # exactly one JSON line, from rank 0, with the contract's keys
    assert line1 == ""
    d = json.loads(line0)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["roofline"]["bound"] == "hbm"
