"""GPU: `python bench.py --gpus 2` with no launcher around it, on the real device path.

A 1-GPU box cannot hold two RCCL ranks, so the run is the rehearsal mode (IIV_BENCH_REHEARSE_ON_ONE_GPU=1: both ranks
bind GPU 0, the process group is gloo): everything else is the multi-GPU path -- the parent that starts the ranks, the
rendezvous on 127.0.0.1, the clip-count agreement, disjoint seeds, the barriers around the timed region, the MAX over
ranks, one JSON line from rank 0.  (The CPU suite runs the same command line with a stand-in for the device work:
tests/test_multiprocess_gloo.py.)"""

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_starts_two_ranks_itself_on_the_device_path():
    env = dict(os.environ, IIV_BENCH_REHEARSE_ON_ONE_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--streams", "512", "--steps", "2",
                        "--warmup", "1", "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout            # one line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and d["launcher"] == "bench.py" and d["dist_backend"] == "gloo"
    assert d["per_rank_frames_per_s"]["ranks"] == 2
    assert d["per_rank_stream_seeds"] == [[1, 512], [513, 1024]]          # disjoint streams per rank
    assert d["config"]["parallelism"].startswith("REHEARSAL")              # and the line says what it is
    # whole job (both ranks' frames) over the max-over-ranks time
    assert abs(d["value"] - 2 * 50 * 512 * 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert d["roofline"]["launches"] > 0 and d["roofline"]["frac"] < 1.0


@pytest.mark.gpu
def test_bench_refuses_more_ranks_than_gpus_on_the_device_path():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "IIV_BENCH_REHEARSE_ON_ONE_GPU"):
        env.pop(k, None)
    import torch
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "visible" in r.stderr and not r.stdout.strip()
