"""bench.GpuBackend's interface without a device (tests only).

bench.main() runs its whole multi-rank control flow -- rank / world logic, clip-count agreement, seeding,
warm-up, timed loop, reductions, rank 0's JSON line, teardown -- against this class on CPU over gloo
(tests/test_multiprocess_gloo.py; `python bench.py --backend bench_standin:CpuStandIn`).  It computes
nothing: it records what it was asked to do.  The product path has no CPU fallback."""

import os

import torch


class CpuStandIn:
    """bench.GpuBackend's interface without a device: records the calls of bench.main()."""

    dist_backend = "gloo"
    is_gpu = False

    def __init__(self, args, local_rank, world):
        import stream_batch
        self.args = args
        self.device = torch.device("cpu")
        self.dhgr = args.mode == "DHGR"
        self.clock = stream_batch.MovieClock(self.dhgr)
        self.rank = int(os.environ.get("RANK", "0"))
        if os.environ.get("IIV_STANDIN_FAIL_RANK") == str(self.rank):
            raise SystemExit(3)    # (test_bench_gpus_flag_starts_the_ranks_itself: a failing rank fails the call)
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
        return {0: torch.zeros((sum(s[3] for s in segs), 6), dtype=torch.uint8)}

    def check(self):
        self.log["checks"] += 1

    def profile(self, on):
        pass

    def profile_read(self):
        return {"prologue_ms": 1.0, "greedy_ms": 2.0, "prologue_launches": 1, "greedy_launches": 1}

    def uses_wave_kernel(self):
        return True

    # what bench.py checks a committed counter run against (tests/test_bench_counters.py)
    def build_id(self):
        return os.environ.get("IIV_STANDIN_BUILD_ID", "standin00000")

    def launch_forms(self):
        return {"plain": 0, "shared": 7, "team": 0, "workgroup": 0}

    def input_stats(self):
        return 0.02, "shared"
