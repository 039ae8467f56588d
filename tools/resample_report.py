#!/usr/bin/env python3
"""Sample-rate conversion on the GPU: batches of files at another rate than the analyser's through
afx_batch_create_from_raw (LoadSample front end + conversion), timed around the call; under rocprofv3 the per-kernel
times (tools/prof_resample.sh).  usage: resample_report.py [n_files] [seconds] [rate ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import afec_amd as afx  # noqa: E402

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rates = [int(v) for v in sys.argv[3:]] or [48000, 96000, 22050]
rng = np.random.default_rng(7)
plan = afx.Plan()
for rate in rates:
    n = int(rate * seconds)
    pool = []
    for _ in range(16):
        t = np.arange(n) / rate
        x = 0.5 * np.sin(2 * np.pi * rng.uniform(80, 2000) * t) + 0.4 * rng.uniform(-1, 1, n) * np.exp(-t * rng.uniform(3, 30))
        x = np.stack([x, 0.8 * np.roll(x, 7)], axis=1)
        pool.append(np.round(x / np.max(np.abs(x)) * 30000).astype(np.int16))
    files = [(pool[i % 16], 2, rate) for i in range(n_files)]
    same = [(pool[i % 16], 2) for i in range(n_files)]
    for what, fs in (("at the analyser's rate", same), (f"at {rate} Hz", files)):
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            batch, infos = plan.batch_from_raw(fs, afx.D_MFCC)
            batch.sync()
            dt = time.perf_counter() - t0
            out_samples = sum(i["n_samples"] for i in infos)
            batch.close()
            best = dt if best is None else min(best, dt)
        print(f"{n_files} stereo files of {seconds} s {what}: create_from_raw {best * 1e3:8.2f} ms  ({n_files / best / 1e3:7.1f} k files/s, "
              f"{out_samples / 1e6:.1f} M samples out)", flush=True)
plan.close()
