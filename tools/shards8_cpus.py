#!/usr/bin/env python3
"""The crawler with 1, 2, 4, 8 shards on ONE device (the G > 1 branch: file i -> shard i mod G, each shard its own analyser,
workers and cursor): files/s and the host CPUs the crawl keeps busy -- what 8 GPUs' worth of host threads cost (the
box's CPU quota is 16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import bench
from afec_amd import hostlib
pool = [bench.wav_image(f, 2) for f in bench.make_c4_files(64, 99)]
images = [pool[i % 64] for i in range(12500)]
for shards, workers in ((1, 8), (2, 4), (4, 2), (8, 1), (8, 2)):
    best = None
    for _ in range(3):
        st = hostlib.crawl(images, devices=(0,) * shards, workers=workers, files_per_batch=512)
        if best is None or st["seconds"] < best["seconds"]:
            best = st
    print(f"{shards} shards x {workers} workers on device 0: {best['files'] / best['seconds'] / 1e3:7.1f} k files/s, "
          f"{best['cpu_seconds'] / best['seconds']:5.2f} busy host CPUs, files per shard {best['files_per_device']}")
