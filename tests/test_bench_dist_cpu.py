"""The N>1 plumbing of bench.py on CPU: two gloo ranks, barrier + max-over-ranks time +
sum-over-ranks frames (replicas only -- the path has no data-path collective)."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import bench
    r, w, local, dist = bench.dist_setup(world)
    assert (r, w, local) == (rank, world, rank) and dist is not None
    dist.barrier()
    seconds, frames = bench.reduce_max_sum(dist, 1.0 + rank, 1000 * (rank + 1))
    q.put((rank, seconds, frames))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_aggregation():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # both ranks see max time = 2.0 s and total frames = 3000
    assert got == [(0, 2.0, 3000.0), (1, 2.0, 3000.0)]


def test_single_process_path_needs_no_torch():
    sys.path.insert(0, ROOT)
    import bench
    old = {k: os.environ.pop(k, None) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    try:
        assert bench.dist_setup(1) == (0, 1, 0, None)
        assert bench.reduce_max_sum(None, 0.5, 10) == (0.5, 10)
    finally:
        for k, v in old.items():
            if v is not None:
                os.environ[k] = v
