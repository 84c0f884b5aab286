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
sys.path.insert(0, os.path.join(ROOT, "tests"))


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


from bench_standin import CpuStandIn as _CpuStandIn  # noqa: E402


def _bench_worker(rank, world, port, q, argv):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"] = str(rank)
    os.environ["LOCAL_RANK"] = str(rank)
    os.environ["WORLD_SIZE"] = str(world)
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "ii-vision_amd", "transcoder")):
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
    made[0].log["affinity"] = sorted(os.sched_getaffinity(0))
    q.put((rank, out["value"], out["config"], made[0].log, buf.getvalue()))


def _fake_sysfs(root):
    """A two-GPU, two-socket machine as bench.pin_to_gpu_numa_node reads it: KFD node 0 is the CPU, nodes 1 and 2 are GPUs
    whose render nodes sit on NUMA nodes 0 and 1, four CPUs each (this container has eight)."""
    def put(path, text):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(text)
    put(root + "/class/kfd/kfd/topology/nodes/0/properties", "cpu_cores_count 8\nsimd_count 0\ndrm_render_minor 0\n")
    for g in (1, 2):
        put(root + "/class/kfd/kfd/topology/nodes/%d/properties" % g, "cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor %d\n" % (127 + g))
        put(root + "/class/drm/renderD%d/device/numa_node" % (127 + g), "%d\n" % (g - 1))
        put(root + "/devices/system/node/node%d/cpulist" % (g - 1), "%d-%d\n" % (4 * (g - 1), 4 * (g - 1) + 3))


def test_bench_main_runs_in_two_ranks(tmp_path, monkeypatch):
    _fake_sysfs(str(tmp_path))
    monkeypatch.setenv("IIV_BENCH_SYSFS", str(tmp_path))   # (the spawned ranks inherit it)
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    argv = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--frames-per-step", "4", "--config", "5"]
    (r0, v0, cfg0, log0, line0), (r1, v1, cfg1, log1, line1) = _spawn(_bench_worker, 2, (argv,))
    # every rank pinned itself to the CPUs of its GPU's NUMA node before anything else, and the line says so
    if len(os.sched_getaffinity(0)) >= 8:
        assert log0["affinity"] == [0, 1, 2, 3] and log1["affinity"] == [4, 5, 6, 7]
        d_ = json.loads(line0)
        assert d_["per_rank_cpu_affinity"]["numa_node"] == [0, 1] and d_["per_rank_cpu_affinity"]["n_cpus"] == [4, 4]
        assert d_["per_rank_cpu_affinity"]["rank0"]["pinned"] and d_["per_rank_cpu_affinity"]["rank0"]["cpus"] == "0,1,2,3"
    # both ranks settled on the clip count the smaller rank can hold (MIN over ranks), and the
    # whole-job value counts both GPUs
    assert cfg0["streams_per_gpu"] == cfg1["streams_per_gpu"] == log0["clips"][0] == log1["clips"][0]
    assert cfg0["streams_per_gpu"] < 12288
    assert v0 == v1 and v0 > 0
    assert cfg0["palette"] == "IIGS" and "IIGS" in cfg0["workload"] and "2 GPU" in cfg0["parallelism"]
    # disjoint seeds and clips per rank; warm-up + events leg + timed steps ran; state was checked on both sides
    assert log0["seeds"][0] != log1["seeds"][0] and log0["clips"][2] != log1["clips"][2]
    assert log0["steps"] == log1["steps"] == 1 + 3 + 3 and log0["checks"] == 3 and log0["sync"] >= 4
    # exactly one JSON line, from rank 0, with the contract's keys
    assert line1 == ""
    d = json.loads(line0)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d
    # (the greedy kernel's HBM fraction is the contract's yardstick; what binds it -- instruction issue -- is labelled as such)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["roofline"]["bound"] == "issue" and d["roofline"]["unit"] == "GB/s"
    assert d["dist_backend"] == "gloo" and d["world_size"] == 2
    assert len(d["per_rank_ms_per_step"]["all"]) == 2 and d["per_rank_ms_per_step"]["min"] <= d["per_rank_ms_per_step"]["max"]
    a, b = d["per_rank_stream_seeds"]
    assert a[1] < b[0] or b[1] < a[0]          # disjoint seed ranges


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2 ...` with NO launcher and NO rank environment: the script starts its two ranks itself
    (bench.launch_ranks), each builds its own process group, and rank 0's one JSON line says what the group saw."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "tests"), os.path.join(ROOT, "ii-vision_amd", "transcoder"), env.get("PYTHONPATH", "")])
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--frames-per-step", "4",
           "--backend", "bench_standin:CpuStandIn"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1                                   # exactly one line, rank 0's
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and d["dist_backend"] == "gloo" and d["launcher"] == "bench.py"
    assert d["per_rank_frames_per_s"]["ranks"] == 2
    (a0, a1), (b0, b1) = d["per_rank_stream_seeds"]
    assert a1 < b0 or b1 < a0                                # disjoint seed ranges => independent streams
    assert "2 GPU" in d["config"]["parallelism"] and d["scaling"] == "weak"
    # --gpus 1: no ranks started, no process group, the line as before
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "0", "--frames-per-step", "4",
                         "--backend", "bench_standin:CpuStandIn", "--streams", "8"], env=dict(env, RANK="0"), capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0, r1.stderr[-2000:]
    d1 = json.loads(r1.stdout.strip().splitlines()[-1])
    assert d1["n_gpus"] == 1 and d1["world_size"] == 1 and d1["dist_backend"] is None and d1["launcher"] is None
    # a rank that fails takes the whole call down with a non-zero exit
    r2 = subprocess.run(cmd + ["--mode", "DHGR", "--streams", "-7"], env=dict(env, IIV_STANDIN_FAIL_RANK="1"), capture_output=True, text=True, timeout=300)
    assert r2.returncode != 0


def test_bench_refuses_more_gpus_than_visible():
    """No GPU in the build container: the real backend with --gpus 2 must refuse before starting anything."""
    import subprocess
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two GPUs visible here")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "visible" in r.stderr and r.stdout.strip() == ""
