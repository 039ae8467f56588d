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


def _run_bench(*argv, env=None, timeout=300):
    import subprocess
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=e, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=timeout)


def test_plain_gpus_n_starts_its_own_ranks_and_a_failed_rank_fails_the_job():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (Crawler.cpp:706-728 one level up).
    Rank 1 exits with status 7 before it joins; rank 0 is then waiting for it in the rendezvous: the launcher must end
    it and return the failing status instead of hanging -- and must not print a line."""
    r = _run_bench("--gpus", "2", "--fail-rank", "1", "--steps", "1", "--warmup", "0", "--buffers", "1", "--no-cpu-baseline", "--no-single")
    assert r.returncode == 7, (r.returncode, r.stderr.decode()[-2000:])
    assert b"rank 1 of 2 exited with status 7" in r.stderr
    assert not any(line.startswith(b"{") for line in r.stdout.splitlines())


def test_without_a_gpu_every_rank_fails_loudly_and_no_line_is_printed():
    """This container has no GPU: the ranks' plans fail (AFX_ERR_NO_DEVICE, no CPU fallback), the job's status is
    non-zero and nothing that looks like a result reaches stdout."""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible: the ranks would run")
    r = _run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--buffers", "1", "--no-cpu-baseline", "--no-single")
    assert r.returncode != 0
    assert b"cannot use HIP device" in r.stderr
    assert not any(line.startswith(b"{") for line in r.stdout.splitlines())


def test_a_world_size_that_contradicts_gpus_is_an_error():
    """a launcher that exported WORLD_SIZE=1 for --gpus 2 used to get a 1-GPU number labelled n_gpus 1 and a warning"""
    r = _run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--buffers", "1", "--no-cpu-baseline", "--no-single",
                   env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and b"--gpus 2 but WORLD_SIZE=1" in r.stderr
    assert not any(line.startswith(b"{") for line in r.stdout.splitlines())


def test_a_launcher_that_is_told_to_stop_takes_its_ranks_with_it():
    """SIGTERM to `bench.py --gpus 2` (a driver's timeout) while its ranks are running: the launcher ends them (their PIDs
    are gone afterwards) and leaves with 128 + SIGTERM -- nothing stays behind on the GPUs."""
    import signal
    import subprocess
    import time
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stall-seconds", "120", "--steps", "1",
                          "--buffers", "1", "--no-cpu-baseline", "--no-single"], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    kids = []
    for _ in range(100):                       # the launcher imports numpy first: wait for its two ranks
        kids = [int(k) for k in subprocess.run(["ps", "-o", "pid=", "--ppid", str(p.pid)], stdout=subprocess.PIPE).stdout.split()]
        if len(kids) == 2:
            break
        time.sleep(0.1)
    assert len(kids) == 2, kids
    p.send_signal(signal.SIGTERM)
    assert p.wait(30) == 128 + signal.SIGTERM
    time.sleep(0.5)

    def running(pid):
        try:
            return open(f"/proc/{pid}/stat").read().rsplit(")", 1)[1].split()[0] != "Z"
        except OSError:
            return False
    assert [k for k in kids if running(k)] == []
