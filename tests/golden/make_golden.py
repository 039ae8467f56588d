#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own compiled objects.

Runs only in the build container (needs /root/reference): `make -C oracle ref` compiles the
reference sources where they lie into oracle/_ref/ref_driver, and this script feeds it the seeded
signals below.  The fixtures are data (inputs + the reference's outputs); no reference source
text is stored.  Record layout: oracle/afx_oracle.h (AFXO_*).

    python tests/golden/make_golden.py
"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
SR, FFT, HOP = 44100, 2048, 1024
NFRAMES = 8
N = FFT + (NFRAMES - 1) * HOP


def signals():
    """name -> float32 signal (float32 so the GPU path can be fed bit-identical PCM)."""
    rng = np.random.default_rng(20261003)
    t = np.arange(N, dtype=np.float64)
    s = {}
    s["sine1k"] = np.sin(2 * np.pi * 1000.0 * t / SR)
    s["sinebin64"] = 0.8 * np.sin(2 * np.pi * 64.0 * t / FFT)
    s["silence"] = np.zeros(N)
    imp = np.zeros(N); imp[1500] = 1.0; imp[6000] = -0.5
    s["impulse"] = imp
    s["noise"] = rng.uniform(-1.0, 1.0, N)
    dur = N / SR
    s["chirp"] = 0.9 * np.sin(2 * np.pi * (20.0 * t / SR + 0.5 * (20000.0 - 20.0) / dur * (t / SR) ** 2))
    s["square220"] = np.sign(np.sin(2 * np.pi * 220.0 * t / SR + 0.1))
    s["quiet"] = 1e-5 * rng.uniform(-1.0, 1.0, N)
    s["dc"] = np.full(N, 0.5)
    env = np.exp(-t / (0.05 * SR))
    s["mix"] = (0.4 * np.sin(2 * np.pi * 110.0 * t / SR) + 0.3 * np.sin(2 * np.pi * 1760.0 * t / SR)
                + 0.25 * env * rng.uniform(-1.0, 1.0, N))
    lead = np.concatenate([np.zeros(2205), 0.7 * np.sin(2 * np.pi * 440.0 * t[:N - 2205] / SR)])
    s["lead_silence"] = lead
    return {k: v.astype(np.float32) for k, v in s.items()}


def long_signals():
    """Longer inputs for the stateful neighbours (whitening follower, pitch): name -> float32 signal."""
    rng = np.random.default_rng(20261004)
    n = FFT + 59 * HOP
    t = np.arange(n, dtype=np.float64)
    s = {}
    # decaying harmonic notes at changing pitches, short gaps of silence in between
    notes = np.zeros(n)
    at = 0
    for f0, dur in [(196.0, 9000), (261.6, 7000), (329.6, 12000), (98.0, 14000), (523.3, 8000), (147.0, 12000)]:
        if at >= n:
            break
        m = min(dur, n - at)
        tt = np.arange(m) / SR
        tone = sum((0.6 / h) * np.sin(2 * np.pi * f0 * h * tt) for h in range(1, 6))
        notes[at:at + m] += 0.5 * np.exp(-tt / 0.12) * tone
        at += dur + 1500
    s["notes"] = notes
    # amplitude-modulated coloured noise
    white = rng.uniform(-1.0, 1.0, n)
    col = np.zeros(n)
    acc = 0.0
    for i in range(n):
        acc = 0.92 * acc + 0.08 * white[i]
        col[i] = acc
    s["amnoise"] = 2.5 * col * (0.55 + 0.45 * np.sin(2 * np.pi * 3.0 * t / SR))
    return {k: v.astype(np.float32) for k, v in s.items()}


def raw_files():
    """Decoded interleaved PCM for the LoadSample front end: name -> (array, channels).  int16, float32, or
    uint8 holding packed little-endian int24."""
    rng = np.random.default_rng(20261005)
    t = np.arange(12000) / SR
    tone = 0.4 * np.sin(2 * np.pi * 330.0 * t) * np.exp(-t / 0.1)
    f = {}
    lead = np.concatenate([np.zeros(700), tone, np.zeros(1300)])
    f["i16_mono_trim"] = (np.round(lead * 20000).astype(np.int16), 1)
    st = np.stack([tone, 0.8 * np.roll(tone, 7)], axis=1)
    f["i16_stereo"] = (np.round(st * 30000).astype(np.int16).reshape(-1), 2)
    v = np.round(0.6 * rng.uniform(-1, 1, 9000) * 8388607).astype(np.int32)
    b = np.zeros((v.size, 3), dtype=np.uint8)
    b[:, 0] = v & 0xFF; b[:, 1] = (v >> 8) & 0xFF; b[:, 2] = (v >> 16) & 0xFF
    f["i24_mono"] = (b.reshape(-1), 1)
    fl = np.stack([1.3 * tone, -0.9 * tone, 0.2 * rng.standard_normal(tone.size)], axis=1)   # clips above 1.0
    f["f32_three_channels_clipping"] = (fl.astype(np.float32).reshape(-1), 3)
    f["i16_short"] = (np.round(0.5 * np.sin(2 * np.pi * 1000 * np.arange(300) / SR) * 32767).astype(np.int16), 1)
    f["i16_all_zero"] = (np.zeros(5000, dtype=np.int16), 1)
    f["i16_quiet"] = (np.round(3.0 * rng.standard_normal(6000)).astype(np.int16), 1)      # amplification >> 1
    f["i16_len_mod_fft_ge_half"] = (np.round(8000 * rng.uniform(-1, 1, 2048 * 2 + 1500)).astype(np.int16), 1)
    return f


def run_ref_load(data, channels):
    data = np.ascontiguousarray(data)
    fmt = {np.dtype(np.int16): 0, np.dtype(np.uint8): 1, np.dtype(np.float32): 2}[data.dtype]
    frames = data.size // (channels * (3 if fmt == 1 else 1))
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("<iiq", fmt, channels, frames))
            f.write(data.tobytes())
        subprocess.check_call([REF, "load", fin, fout])
        raw = open(fout, "rb").read()
    peak, rms, off, lead, trail, _, n = struct.unpack("<ffiiiiq", raw[:32])
    x = np.frombuffer(raw[32:], dtype=np.float64).copy()
    assert x.size == n
    return np.array([peak, rms], dtype=np.float32), np.array([off, lead, trail, n], dtype=np.int64), x


def column_values(n, width=0, salt=0):
    """Deterministic, exactly representable test values for the column encoder (the C++ host test generates the
    same ones): v[i] = ((i * i + 7 * salt) % 97) / 8 - 3, with a zero, a negative zero and an integer mixed in."""
    m = n * max(width, 1)
    i = np.arange(m, dtype=np.int64)
    v = ((i * i + 7 * salt) % 97) / 8.0 - 3.0
    if m > 2:
        v[1] = 0.0
        v[2] = -0.0
    if m > 3:
        v[3] = 43.0 * 1000
    return v.reshape(n, width) if width else v


def run_ref_msgpack(values, width):
    values = np.ascontiguousarray(values, dtype=np.float64)
    rows = values.shape[0]
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("<iiq", width, 0, rows))
            f.write(values.tobytes())
        subprocess.check_call([REF, "msgpack", fin, fout])
        return np.frombuffer(open(fout, "rb").read(), dtype=np.uint8).copy()


def run_ref(bufs, cap=0, mode="frames", record=1147):
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("<q", len(bufs)))
            for b in bufs:
                b = np.asarray(b, dtype=np.float64)
                f.write(struct.pack("<q", b.size))
                f.write(b.tobytes())
        subprocess.check_call([REF, mode, fin, fout, str(int(cap))])
        raw = open(fout, "rb").read()
    n = struct.unpack("<q", raw[:8])[0]
    return np.frombuffer(raw[8:], dtype=np.float64).reshape(n, record).copy()


def main():
    if not os.path.exists(REF):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    out = {}
    for name, x in signals().items():
        rec = run_ref([x.astype(np.float64)])
        assert rec.shape == (NFRAMES, 1147), rec.shape
        out["in_" + name] = x
        out["ref_" + name] = rec
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "frames.npz"), **out)

    # stateful neighbours: 11 scalars + whitened spectrum per frame; the long signals keep
    # the scalars only
    out = {}
    for name, x in signals().items():
        rec = run_ref([x.astype(np.float64)], mode="neighbours", record=1035)
        assert rec.shape == (NFRAMES, 1035), rec.shape
        out["ref_" + name] = rec
    for name, x in long_signals().items():
        rec = run_ref([x.astype(np.float64)], mode="neighbours", record=1035)
        out["in_" + name] = x
        out["ref_" + name] = rec[:, :11].copy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "neighbours.npz"), **out)

    # LoadSample front end
    out = {}
    for name, (data, channels) in raw_files().items():
        pr, info, x = run_ref_load(data, channels)
        out["raw_" + name] = data
        out["channels_" + name] = np.array(channels)
        out["peakrms_" + name] = pr
        out["info_" + name] = info
        out["data_" + name] = x
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "load.npz"), **out)

    # effective lengths (per file): the golden signals, the long ones, and a few shaped for the three floors
    rng = np.random.default_rng(20261006)
    eff_in = dict(signals())
    eff_in.update(long_signals())
    ramp = np.concatenate([np.zeros(3000), np.linspace(0, 1, 20000), np.linspace(1, 0, 30000) ** 3, np.zeros(4000)])
    eff_in["ramps"] = (ramp * np.sin(2 * np.pi * 300 * np.arange(ramp.size) / SR)).astype(np.float32)
    eff_in["one_spike"] = np.zeros(5000, dtype=np.float32); eff_in["one_spike"][1234] = 0.5
    eff_in["just_below_24dB"] = (0.06 * rng.uniform(-1, 1, 9000)).astype(np.float32)
    out = {}
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        names = sorted(eff_in)
        with open(fin, "wb") as f:
            f.write(struct.pack("<q", len(names)))
            for k in names:
                x = eff_in[k].astype(np.float64)
                f.write(struct.pack("<q", x.size)); f.write(x.tobytes())
        subprocess.check_call([REF, "efflen", fin, fout])
        res = np.fromfile(fout, dtype=np.float64).reshape(len(names), 3)
    stored_elsewhere = set(signals()) | set(long_signals())      # inputs already in frames.npz / neighbours.npz
    for i, k in enumerate(names):
        if k not in stored_elsewhere:
            out["in_" + k] = eff_in[k]
        out["ref_" + k] = res[i]
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "efflen.npz"), **out)

    # BLOB encoding of the low-level columns (the reference's vendored msgpack-c through SToMsgpack's calls)
    import hashlib
    out = {}
    for n in (0, 1, 15, 16, 860):
        out[f"vr_{n}"] = run_ref_msgpack(column_values(n, 0, n), 0)
    for rows, width in ((0, 14), (3, 14), (20, 28), (860, 14)):
        out[f"vvr_{rows}x{width}"] = run_ref_msgpack(column_values(rows, width, rows + width), width)
    big = run_ref_msgpack(column_values(70000, 0, 5), 0)                 # array 32 header; only its digest is kept
    out["vr_70000_sha256"] = np.frombuffer(hashlib.sha256(big.tobytes()).digest(), dtype=np.uint8).copy()
    out["vr_70000_head"] = big[:14].copy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "columns.npz"), **out)

    # tables
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "t.bin")
        subprocess.check_call([REF, "tables", p], stdout=subprocess.DEVNULL)
        tab = np.fromfile(p, dtype=np.float64)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "tables.npz"),
                        window=tab[:FFT], mel=tab[FFT:].reshape(14, FFT // 2))

    # frame-count rule incl. the 20 s cap (SampleAnalyser.cpp:760-764, 814)
    rows = []
    for n, cap in [(0, 0), (1, 0), (2047, 0), (2048, 0), (2049, 0), (3071, 0), (3072, 0), (4096, 0),
                   (882000, 1), (882001, 1), (900000, 1), (900000, 0), (883712, 1)]:
        rec = run_ref([np.zeros(n)], cap)
        rows.append((n, cap, rec.shape[0]))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "framecount.npz"),
                        rows=np.array(rows, dtype=np.int64))
    print("wrote tests/golden/{frames,neighbours,load,efflen,columns,tables,framecount}.npz")


if __name__ == "__main__":
    sys.exit(main())
