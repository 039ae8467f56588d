#!/usr/bin/env python3
"""Streaming host driver: warm files/s over workers x files per batch, one process (the crawler persists).
usage: e2e_sweep.py [n_files] [W:B ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from afec_amd import hostlib  # noqa: E402

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
configs = [tuple(int(v) for v in s.split(":")) for s in sys.argv[2:]] or [
    (3, 256), (4, 256), (2, 512), (3, 512), (4, 512), (6, 512), (2, 1024), (3, 1024), (4, 1024), (6, 1024), (8, 1024),
    (3, 2048), (4, 2048), (6, 2048)]
files = bench.make_c4_files(64, 1234)
pool = [bench.wav_image(f, 2) for f in files]
images = [pool[i % len(pool)] for i in range(n_files)]
hostlib.crawl(images[:4096], workers=3, files_per_batch=256)   # set-up
def throttled():
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0))
    except OSError:
        return 0, 0


for w, b in configs:
    best = None
    for _ in range(3):
        c0, th0 = os.times(), throttled()
        st = hostlib.crawl(images, workers=w, files_per_batch=b)
        c1, th1 = os.times(), throttled()
        st["cpus"] = (c1.user + c1.system - c0.user - c0.system) / max(c1.elapsed - c0.elapsed, 1e-9)
        st["throttled_ms"] = (th1[1] - th0[1]) / 1e3
        if best is None or st["seconds"] < best["seconds"]:
            best = st
    print(f"{w} x {b:5d}: {best['seconds'] * 1e3:7.1f} ms  {n_files / best['seconds'] / 1e3:6.1f} k files/s  "
          f"{best['pcm_bytes'] / best['seconds'] / 1e9:5.1f} GB/s up  {best['cpus']:4.1f} CPUs busy, throttled {best['throttled_ms']:.1f} ms", flush=True)
