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
