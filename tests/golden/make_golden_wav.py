#!/usr/bin/env python3
"""Goldens for the WAV path (first clause): WAV files of every sample type the reference's TWaveFile
accepts (8-bit unsigned, 16 / 24 / 32-bit integer, 32 / 64-bit float; WaveFile.cpp:14-47), and what LoadSample makes
of their data chunks according to the reference's own converters (oracle/_ref/ref_driver load: TSampleConverter,
TMathT, TAudioMath).  Run in the build container (needs oracle/_ref); writes tests/golden/load_wav.npz."""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
sys.path.insert(0, ROOT)

from tests._wav import wav_bytes  # noqa: E402


def run_ref_load(payload, fmt, channels, frames):
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("<iiq", fmt, channels, frames))
            f.write(payload)
        subprocess.check_call([REF, "load", fin, fout])
        raw = open(fout, "rb").read()
    peak, rms, off, lead, trail, _, n = struct.unpack("<ffiiiiq", raw[:32])
    x = np.frombuffer(raw[32:], dtype=np.float64).copy()
    assert x.size == n
    return np.array([peak, rms], dtype=np.float32), np.array([off, lead, trail, n], dtype=np.int64), x


def cases():
    rng = np.random.default_rng(77)
    n = 9000
    t = np.arange(n)
    tone = 0.6 * np.sin(2 * np.pi * 523.25 * t / 44100) * np.exp(-t / 5000.0)
    tone[:300] = 0.0
    noise = 0.3 * rng.uniform(-1, 1, n) * np.linspace(1, 0.01, n)
    stereo = np.stack([tone + noise, 0.8 * np.roll(tone, 7)], axis=1)
    out = {}
    out["u8_mono"] = (np.clip(np.round(tone * 127 + 128), 0, 255).astype(np.uint8), 1, 8, False)
    out["i16_stereo"] = (np.round(stereo * 30000).astype(np.int16), 2, 16, False)
    i24 = np.round((tone + noise) * 8000000).astype(np.int32)
    out["i24_mono"] = (i24, 1, 24, False)
    hot = np.round(stereo * 2147483000.0 * 1.2)               # clips: exercises the [-32768, 32767] clamp
    out["i32_stereo"] = (np.clip(hot, -2147483648, 2147483647).astype(np.int32), 2, 32, False)
    out["f32_mono"] = ((1.3 * (tone + noise)).astype(np.float32), 1, 32, True)   # beyond [-1, 1]: clamped
    out["f64_stereo"] = (stereo.astype(np.float64), 2, 64, True)
    return out


def main():
    res = {}
    for name, (data, channels, bits, is_float) in cases().items():
        image = wav_bytes(data, channels, bits, is_float, extra_chunks=(name in ("i24_mono", "f64_stereo")),
                          extensible=(name == "f32_mono"))
        if bits == 24:
            payload = b"".join(struct.pack("<i", int(v))[:3] for v in data.reshape(-1))
            fmt = 1
        else:
            payload = np.ascontiguousarray(data).tobytes()
            fmt = {8: 5, 16: 0, 32: 2 if is_float else 3, 64: 4}[bits]
        frames = data.reshape(-1).size // channels
        pr, info, x = run_ref_load(payload, fmt, channels, frames)
        res["wav_" + name] = np.frombuffer(image, dtype=np.uint8)
        res["props_" + name] = np.array([channels, 44100, bits, frames], dtype=np.int64)
        res["peakrms_" + name] = pr
        res["info_" + name] = info
        res["data_" + name] = x
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "load_wav.npz"), **res)
    print("wrote tests/golden/load_wav.npz:", ", ".join(sorted(k[4:] for k in res if k.startswith("wav_"))))


if __name__ == "__main__":
    main()
