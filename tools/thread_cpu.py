#!/usr/bin/env python3
"""Which threads of the process burn the host CPU during a crawl: per-thread user + system time (/proc/self/task) over
N warm crawls of the C4 share, by thread name -- the crawler's workers and writer are named by the C++ layer's threads
(python: the calling thread), the HIP runtime's own threads keep the names the runtime gives them.
usage: thread_cpu.py [n_files] [workers] [crawls]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import bench  # noqa: E402
from afec_amd import hostlib  # noqa: E402


def threads():
    out = {}
    hz = os.sysconf("SC_CLK_TCK")
    for tid in os.listdir("/proc/self/task"):
        try:
            comm = open(f"/proc/self/task/{tid}/comm").read().strip()
            f = open(f"/proc/self/task/{tid}/stat").read().rsplit(")", 1)[1].split()
            out[tid] = (comm, (int(f[11]) + int(f[12])) / hz)
        except OSError:
            pass
    return out


n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
crawls = int(sys.argv[3]) if len(sys.argv) > 3 else 20
files = bench.make_c4_files(64, 1234)
pool = [bench.wav_image(f, 2) for f in files]
images = [pool[i % len(pool)] for i in range(n_files)]
hostlib.crawl(images, workers=workers, files_per_batch=512)   # set-up
before = threads()
t0 = time.time()
busy = []
for _ in range(crawls):
    st = hostlib.crawl(images, workers=workers, files_per_batch=512)
    busy.append((st["cpu_seconds"] / st["seconds"], n_files / st["seconds"]))
wall = time.time() - t0
after = threads()
rows = {}
for tid, (comm, t) in after.items():
    d = t - before.get(tid, (comm, 0.0))[1]
    rows.setdefault(comm, [0, 0.0])
    rows[comm][0] += 1
    rows[comm][1] += d
crawl_s = sum(n_files / r for _, r in busy)
print(f"{crawls} crawls of {n_files} files, {workers} workers: median {sorted(b for b, _ in busy)[len(busy) // 2]:.2f} busy CPUs, "
      f"median {sorted(r for _, r in busy)[len(busy) // 2] / 1e3:.1f} k files/s; crawl time {crawl_s * 1e3:.0f} ms of {wall * 1e3:.0f} ms wall")
top = sorted(((after[t][1] - before.get(t, ("", 0.0))[1], t, after[t][0]) for t in after), reverse=True)[:8]
print("  busiest threads that outlive a crawl: " + ", ".join(f"{c}[{t}] {d * 1e3:.0f} ms" for d, t, c in top if d > 0))
wchan = {}
for d, t, c in top[:4]:
    try:
        wchan[t] = open(f"/proc/self/task/{t}/wchan").read().strip()
    except OSError:
        pass
print("  where they sleep:", wchan)
for comm, (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    if t > 0:
        print(f"  {comm:20s} x{n:3d}  {t * 1e3:8.0f} ms CPU  = {t / crawl_s:5.2f} CPUs while crawling")
