#!/usr/bin/env python3
"""Generate tests/golden/rhythm.npz from the REFERENCE's own compiled objects (build container only).

`make -C oracle ref` builds oracle/_ref/ref_driver from the reference sources where they lie; its modes
  onsetfft   TFftWindow::SFillBuffer + ooura_cdft + TAudioMath::Magnitude / Phase  (the onset STFT front end of
             TOnsetFftProcessor::LoadFrame, OnsetDetector.cpp:116-160)
  beattrack  aubio's beattracking.c as TRhythmTracker::CalculateTempo drives it (RhythmTracker.cpp:176-197)
produce the expected values stored here next to their inputs.  TOnsetDetector / TRhythmTracker / TCannyWindow do not
link in this image (see oracle/Makefile), so the detector and the heuristics have no reference-generated golden: that
part of the oracle is "parity unpinned" and is covered by hand-computed cases in tests/test_oracle_rhythm.py.

    python tests/golden/make_golden_rhythm.py
"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
SR = 44100


def run_ref(mode, bufs):
    with tempfile.TemporaryDirectory() as d:
        fi, fo = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fi, "wb") as f:
            f.write(struct.pack("<q", len(bufs)))
            for b in bufs:
                b = np.ascontiguousarray(b, dtype=np.float64)
                f.write(struct.pack("<q", b.size))
                f.write(b.tobytes())
        subprocess.check_call([REF, mode, fi, fo])
        return open(fo, "rb").read()


def drum_loop(bpm, seconds, seed, bass=False):
    rng = np.random.default_rng(seed)
    n = int(SR * seconds)
    x = np.zeros(n)
    step = int(round(60.0 / bpm * SR / 2))            # eighth notes
    for k, at in enumerate(range(0, n, step)):
        length = min(3000, n - at)
        t = np.arange(length)
        env = np.exp(-t / 400.0)
        if k % 4 == 0:
            x[at:at + length] += 0.9 * env * np.sin(2 * np.pi * 70.0 * t / SR)                  # kick
        elif k % 4 == 2:
            x[at:at + length] += 0.6 * env * rng.uniform(-1, 1, length)                         # snare
        else:
            x[at:at + length] += 0.25 * np.exp(-t / 120.0) * rng.uniform(-1, 1, length)         # hat
    if bass:
        t = np.arange(n)
        x += 0.2 * np.sin(2 * np.pi * 55.0 * t / SR) * (0.5 + 0.5 * np.sign(np.sin(2 * np.pi * (bpm / 60.0) * t / SR)))
    return x


def signals():
    s = {}
    s["loop120"] = drum_loop(120.0, 4.0, 11)
    s["loop95"] = drum_loop(95.0, 5.05, 12, bass=True)
    t = np.arange(int(0.8 * SR))
    s["oneshot"] = np.exp(-t / 6000.0) * np.sin(2 * np.pi * 330.0 * t / SR)
    mel = np.zeros(int(3.0 * SR))
    at = 0
    for f0, dur in [(220.0, 9000), (277.2, 12000), (329.6, 7000), (440.0, 15000), (196.0, 11000), (246.9, 20000), (293.7, 30000)]:
        length = min(dur, mel.size - at)
        if length <= 0:
            break
        tt = np.arange(length)
        mel[at:at + length] += np.exp(-tt / 5000.0) * (np.sin(2 * np.pi * f0 * tt / SR) + 0.4 * np.sin(2 * np.pi * 2 * f0 * tt / SR))
        at += dur
    s["melody"] = mel
    out = {}
    for k, v in s.items():
        v = v / np.max(np.abs(v))
        out[k] = np.round(v * 32767.0).astype(np.int16)     # stored as 16-bit PCM; analysed as pcm / 32768
    return out


def main():
    if not os.path.exists(REF):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    sig = signals()
    o = _oracle.Oracle()
    data = {}
    series = []
    for name, pcm in sig.items():
        x = pcm.astype(np.float64) / 32768.0
        raw = run_ref("onsetfft", [x])
        frames = struct.unpack("<q", raw[:8])[0]
        polar = np.frombuffer(raw[8:], dtype=np.float32).reshape(frames, 512)
        sel = np.unique(np.concatenate([np.arange(min(48, frames)), np.arange(0, frames, 32), [frames - 1]]))
        data[f"pcm_{name}"] = pcm
        data[f"polar_frames_{name}"] = sel.astype(np.int32)
        data[f"polar_{name}"] = polar[sel]
        r = o.run_rhythm(x)
        for t in range(2):
            series.append((f"{name}_{t}", r["sharpened"][t].copy()))
    # beat tracking on the oracle's sharpened onset series of every signal, plus synthetic series
    rng = np.random.default_rng(77)
    series.append(("noise300", np.abs(rng.normal(size=300))))
    series.append(("halfnormal1000", np.maximum(0.0, rng.normal(size=1000))))
    imp = np.zeros(700)
    imp[::43] = 3.0
    series.append(("impulses43", imp))
    series.append(("short40", np.maximum(0.0, rng.normal(size=40))))
    bt = np.frombuffer(run_ref("beattrack", [s for _, s in series]), dtype=np.float64).reshape(-1, 2)
    data["beat_names"] = np.array([n for n, _ in series])
    for (n, s), row in zip(series, bt):
        data[f"beat_in_{n}"] = s
    data["beat_out"] = bt
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "rhythm.npz"), **data)
    print("wrote rhythm.npz:", {k: v.shape for k, v in data.items() if k.startswith("polar_") or k == "beat_out"})


if __name__ == "__main__":
    main()
