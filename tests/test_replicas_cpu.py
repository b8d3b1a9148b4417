"""N>1 path of bench.py (replicas only, no data-path collective): world_size-2 gloo run on CPU."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import bench
    import torch.distributed as dist
    r, w = bench.dist_setup("gloo")
    assert (r, w) == (rank, world)
    dist.barrier()
    # rank 1 is "slower": the whole-job rate must use the max time and count both ranks' units
    rate, wall_max = bench.replica_aggregate(1.0 + rank, 10.0, w, "cpu")
    q.put((rank, rate, wall_max))
    dist.barrier()
    dist.destroy_process_group()


def test_two_replica_aggregate_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, rate, wall_max in res:
        assert wall_max == pytest.approx(2.0)
        assert rate == pytest.approx(2 * 10.0 / 2.0)


def test_single_process_aggregate():
    sys.path.insert(0, ROOT)
    import bench
    rate, wall = bench.replica_aggregate(0.5, 4.0, 1)
    assert rate == pytest.approx(8.0) and wall == pytest.approx(0.5)


def test_bench_gpus_2_starts_two_ranks():
    """`python bench.py --gpus 2` without a launcher starts the two ranks itself (here: the CPU stub workload under gloo, which runs the
    same launcher, window timing and max-over-ranks aggregation code as the GPU path) and rank 0 prints ONE line with n_gpus = 2."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub-workload", "--steps", "5", "--warmup", "1",
                        "--windows", "3", "--preroll-ms", "20"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 5 and res["config"]["parallelism"] == "replicas x2"
    assert res["data"].startswith("STUB")                  # a stub line can never be mistaken for a measurement
    assert 0.9 < res["ms_per_step"] < 20.0                 # 5 sleeps of 1 ms per window
    assert res["preroll_ms"] >= 20.0


def test_bench_launcher_env_is_one_rank():
    """Under a launcher (WORLD_SIZE set) the process is ONE rank and must not spawn."""
    import json
    import subprocess
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--stub-workload", "--steps", "3", "--warmup", "0",
                        "--windows", "1", "--preroll-ms", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert res["n_gpus"] == 1
