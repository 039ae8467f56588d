#!/usr/bin/env python3
"""Streaming host driver on files of other lengths than C4's one second: mono 16-bit files of 2 s (C3), 5 s, 20 s and a
mix, warm crawler, default options (8 workers, 512 files or 128 MiB per batch)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from afec_amd import hostlib  # noqa: E402

base = bench.make_c3_files(32, 77)                      # 2.0 s each


def files_of(seconds):
    reps = int(np.ceil(seconds / 2.0))
    return [bench.wav_image(np.tile(x, reps)[: int(seconds * 44100)], 1) for x in base]


pools = {2.0: files_of(2.0), 5.0: files_of(5.0), 20.0: files_of(20.0)}
cases = [("2 s mono (C3)", [2.0], 20000), ("5 s mono", [5.0], 8000), ("20 s mono", [20.0], 2000), ("mix 2 / 5 / 20 s", [2.0, 5.0, 20.0, 2.0, 2.0], 10000)]
hostlib.crawl(pools[2.0] * 64)
for name, kinds, n in cases:
    images = [pools[kinds[i % len(kinds)]][(i // len(kinds)) % 32] for i in range(n)]
    best = None
    for _ in range(3):
        st = hostlib.crawl(images)
        if best is None or st["seconds"] < best["seconds"]:
            best = st
    print(f"{name:18s} {n:6d} files in {int(best['batches']):4d} batches: {best['seconds'] * 1e3:7.1f} ms  {n / best['seconds'] / 1e3:6.1f} k files/s  "
          f"{best['frames'] / best['seconds'] / 1e6:5.2f} M frames/s  {best['pcm_bytes'] / best['seconds'] / 1e9:5.1f} GB/s up", flush=True)
