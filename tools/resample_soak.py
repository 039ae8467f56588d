#!/usr/bin/env python3
"""Soak of the sample-rate conversion on the GPU: batches of files with random rates (3 kHz .. 700 kHz, the common ones
more often), lengths (1 sample .. 400 k), channel counts and sample types through afx_batch_create_from_raw; the
normalised samples in the analysis arena must equal the oracle's LoadSample bit for bit.
usage: resample_soak.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import afec_amd as afx  # noqa: E402
from tests import _oracle  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 41
rng = np.random.default_rng(seed)
COMMON = [8000, 11025, 16000, 22050, 24000, 32000, 44056, 47250, 48000, 50000, 64000, 88200, 96000, 176400, 192000, 352800, 384000]
plan = afx.Plan(max_analysis_ms=0)
t_end = time.time() + seconds
files_done = samples_done = batches = 0
while time.time() < t_end:
    files = []
    for _ in range(int(rng.integers(1, 40))):
        rate = int(rng.choice(COMMON)) if rng.random() < 0.7 else int(rng.integers(3000, 700000))
        n = int(rng.choice([1, 2, 3, 17, 4000, 4039, 4040, 4041, 4096])) if rng.random() < 0.25 else int(np.exp(rng.uniform(np.log(10), np.log(400000))))
        ch = int(rng.choice([1, 1, 2, 2, 3, 6]))
        x = rng.uniform(-1, 1, (n, ch)) * np.exp(-np.arange(n)[:, None] / rng.uniform(50, 50000))
        if rng.random() < 0.3:
            x[: int(rng.integers(0, n))] = 0.0
        kind = int(rng.integers(0, 5))
        if kind == 0:
            d = np.round(x * 32767).astype(np.int16)
        elif kind == 1:
            v = np.round(x * 8388607).astype(np.int64).reshape(-1)
            d = np.frombuffer(b"".join(int(k).to_bytes(4, "little", signed=True)[:3] for k in v), dtype=np.uint8).copy() if v.size < 30000 else np.round(x * 32767).astype(np.int16)
        elif kind == 2:
            d = x.astype(np.float32)
        elif kind == 3:
            d = np.round(x * 2147483647).astype(np.int64).astype(np.int32)
        else:
            d = x.astype(np.float64)
        files.append((d, ch, rate))
    batch, infos = plan.batch_from_raw(files, afx.D_MFCC)
    batch.run()
    status = batch.fetch()["buf_status"].tolist()
    for i, (d, ch, rate) in enumerate(files):
        factor = 44100.0 / rate
        if factor < 1.0 / 16.0:
            assert status[i] == -2, (rate, status[i])
            continue
        assert status[i] == 0, (rate, d.shape, status[i])
        want, winfo = _oracle.load_sample(d, ch, file_rate=rate)
        for k in ("data_offset", "silent_leading", "silent_trailing", "n_samples"):
            assert infos[i][k] == winfo[k], (rate, d.shape, ch, k, infos[i][k], winfo[k])
        nf = (want.size - 2048) // 1024 + 1
        kept = (nf - 1) * 1024 + 2048
        got = batch.fetch_samples(i, kept)
        if not np.array_equal(got.view(np.uint64), want[:kept].view(np.uint64)):
            bad = np.nonzero(got != want[:kept])[0]
            print(f"MISMATCH rate {rate} frames {d.shape} channels {ch} dtype {d.dtype}: {bad.size} samples, first {bad[:5]}", flush=True)
            sys.exit(1)
        files_done += 1
        samples_done += want.size
    batch.close()
    batches += 1
print(f"resample soak: {batches} batches, {files_done} files, {samples_done / 1e6:.1f} M normalised samples compared bit for bit in {seconds:.0f} s, seed {seed}: all equal")
