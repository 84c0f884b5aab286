"""CPU, world_size 2 over gloo: the N>1 path of bench.py shards streams across ranks
with no data-path collective; the only cross-rank traffic is the barrier and the
MAX-reduce of the elapsed time."""

import os
import sys

import pytest
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


def test_two_ranks_shard_streams_and_reduce_time():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29511 + os.getpid() % 200
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, s0, e0, d0), (r1, s1, e1, d1) = res
    assert e0 == e1 == 2.0                       # MAX over ranks
    assert not set(s0) & set(s1)                  # disjoint RNG seeds => independent streams
    assert d0 != d1                               # different synthetic clips per rank
