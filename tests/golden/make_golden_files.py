#!/usr/bin/env python3
"""Whole-file goldens at BASELINE's C3 / C4 shapes, chained through the reference's own objects: decoded PCM ->
ref_driver load (TSampleConverter, TMathT, TAudioMath) -> ref_driver frames (LibXtract, Ooura FFT, TAudioMath,
TStatistics) + ref_driver neighbours (aubio, TEnvelopeDetector, TAutocorrelation) on the loaded buffer with the 20 s
cap.  Pins the end-to-end pipeline (LoadSample front end + every per-frame descriptor) to the reference, not only to the
oracle.  Stored: the PCM, LoadSample's offsets, and per frame the 123 spectral + 11 neighbour descriptors (not the
magnitudes).  Run in the build container; writes tests/golden/files.npz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.golden.make_golden import run_ref, run_ref_load  # noqa: E402


def synth(rng, seconds, stereo):
    n = int(44100 * seconds)
    t = np.arange(n) / 44100.0
    x = np.zeros(n)
    for _ in range(int(rng.integers(1, 4))):
        x += rng.uniform(0.2, 0.6) * np.sin(2 * np.pi * rng.uniform(110.0, 4000.0) * t + rng.uniform(0, 6.28))
    x += rng.uniform(0.2, 0.8) * rng.uniform(-1, 1, n) * np.exp(-t / rng.uniform(0.05, 0.5))
    x[:2205] = 0.0                                    # 50 ms of leading silence
    x *= rng.uniform(0.1, 0.9) / np.max(np.abs(x))
    if stereo:                                        # C4: R = L delayed 7 samples x 0.8
        return np.round(np.stack([x, 0.8 * np.roll(x, 7)], axis=1) * 32767).astype(np.int16).reshape(-1), 2
    return np.round(x * 32767).astype(np.int16), 1


def main():
    rng = np.random.default_rng(20261101)
    out = {}
    cases = {"c3_mono_2s_a": (2.0, False), "c3_mono_2s_b": (2.0, False), "c4_stereo_1s_a": (1.0, True), "c4_stereo_1s_b": (1.0, True)}
    for name, (seconds, stereo) in cases.items():
        data, ch = synth(rng, seconds, stereo)
        pr, info, x = run_ref_load(data, ch)
        rec = run_ref([x], cap=1)
        nei = run_ref([x], cap=1, mode="neighbours", record=1035)
        assert rec.shape[0] == nei.shape[0]
        out["raw_" + name] = data
        out["channels_" + name] = np.array(ch)
        out["peakrms_" + name] = pr
        out["info_" + name] = info
        out["spectral_" + name] = rec[:, 1024:].copy()      # the 123 descriptors of oracle/afx_oracle.h (record minus magnitudes)
        out["neighbours_" + name] = nei[:, :11].copy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "files.npz"), **out)
    print("wrote tests/golden/files.npz:", ", ".join(cases), {k: out["spectral_" + k].shape for k in cases})


if __name__ == "__main__":
    main()
