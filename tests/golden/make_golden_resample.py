#!/usr/bin/env python3
"""Goldens of the sample-rate conversion in LoadSample (SampleAnalyser.cpp:563-607) from the reference's own libresample
(3rdParty/Resample/Dist/src, compiled into oracle/_ref/ref_driver by `make -C oracle ref`; mode `resample` drives it
with the call sequence of those lines).

Inputs are seeded (the test regenerates them: `signal` below), outputs are stored: in full for the short cases, as
(sha256, first / last 64 samples) for the long ones.  Run in the build container:
    python tests/golden/make_golden_resample.py      -> tests/golden/resample.npz"""
import hashlib
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
RATES = [48000, 96000, 88200, 22050, 32000, 11025, 8000, 192000, 44099, 44101, 16000, 37800]
SHORT = [1, 2, 17, 37, 1000, 4039, 4040, 4041, 4096, 8200]
LONG = {48000: 60000, 22050: 30000, 96000: 120000, 8000: 9000}


def signal(n, seed):
    """the mono "16-bit float" buffer LoadSample holds in front of the conversion: a drum-like burst, +-32768 range"""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 48000.0
    x = 0.6 * np.sin(2 * np.pi * rng.uniform(60, 4000) * t + rng.uniform(0, 6.28)) + 0.4 * rng.uniform(-1, 1, n) * np.exp(-t * rng.uniform(5, 80))
    return np.round(x * 30000.0).astype(np.float32)


def run_ref_resample(x, file_rate, rate=44100):
    with tempfile.TemporaryDirectory() as td:
        fi, fo = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fi, "wb") as f:
            f.write(struct.pack("<iiq", file_rate, rate, x.size))
            f.write(np.ascontiguousarray(x, np.float32).tobytes())
        subprocess.check_call([REF, "resample", fi, fo])
        raw = open(fo, "rb").read()
    new_size, written, used = struct.unpack("<qqq", raw[:24])
    out = np.frombuffer(raw[24:], dtype=np.float32).copy()
    assert out.size == new_size
    return out, written, used


def main():
    if not os.path.exists(REF):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    out = {}
    cases = []
    for r in RATES:
        for n in SHORT + ([LONG[r]] if r in LONG else []):
            seed = 1000 * (RATES.index(r) + 1) + n % 997
            y, written, used = run_ref_resample(signal(n, seed), r)
            key = f"{r}_{n}"
            cases.append((r, n, seed, y.size, written, used))
            if n <= 8200:
                out["out_" + key] = y
            else:
                out["head_" + key], out["tail_" + key] = y[:64], y[-64:]
                out["sha_" + key] = np.frombuffer(hashlib.sha256(y.tobytes()).digest(), dtype=np.uint8)
    out["cases"] = np.array(cases, dtype=np.int64)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "resample.npz"), **out)
    print(len(cases), "cases;", "written != NewSize:", [(c[0], c[1], c[3], c[4]) for c in cases if c[3] != c[4]][:20],
          "; used != n:", [(c[0], c[1], c[5]) for c in cases if c[5] != c[1]][:20])


if __name__ == "__main__":
    sys.exit(main())
