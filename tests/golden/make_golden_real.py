#!/usr/bin/env python3
"""Real-audio goldens from the reference's own test fixtures (UnitTests.cpp:152-423 crawls these folders).

tests/golden/wav/ holds eleven of the WAVs under Source/Crawler/XUnitTests/Resources/Kicks-vs-Snare-{Train,Test}
(data files, byte for byte): 16- and 24-bit, mono and stereo, one with extra RIFF chunks, the near-silent
"_empty wave file.wav", a file name outside ASCII, and "_Not A Wavefile.wav".  For every readable one this script
chains the decoded PCM through the reference's own objects (oracle/_ref/ref_driver, built from the reference sources
by `make -C oracle ref`):

  load        TSampleConverter / TMathT / TAudioMath   -> the normalised, trimmed, padded buffer + offsets
  frames      LibXtract, Ooura FFT, TStatistics         -> the 123 spectral descriptors per frame
  neighbours  aubio, TEnvelopeDetector, TAutocorrelation-> the 11 neighbour descriptors per frame
  onsetfft    TFftWindow + ooura_cdft + Magnitude/Phase -> the rhythm tracker's polar spectra (a selection of frames)
  beattrack   aubio beattracking.c                      -> tempo / confidence of the oracle's sharpened onset series

and stores the expected values in tests/golden/real.npz.  Run in the build container:  python tests/golden/make_golden_real.py"""
import glob
import os
import struct
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tests.golden.make_golden import run_ref  # noqa: E402
from tests.golden.make_golden_wav import run_ref_load  # noqa: E402
from tests.golden.make_golden_rhythm import run_ref as run_ref_rhythm  # noqa: E402
from tests._wav import parse_wav  # noqa: E402
import _oracle  # noqa: E402

WAV_DIR = os.path.join(ROOT, "tests", "golden", "wav")
REF = os.path.join(ROOT, "oracle", "_ref", "ref_driver")


def main():
    if not os.path.exists(REF):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    o = _oracle.Oracle()
    out, names, series = {}, [], []
    for path in sorted(glob.glob(os.path.join(WAV_DIR, "*.wav"))):
        name = os.path.basename(path)
        image = open(path, "rb").read()
        try:
            channels, rate, bits, frames, payload = parse_wav(image)
        except ValueError:
            continue                                           # "_Not A Wavefile.wav": no golden, the reader must reject it
        fmt = {16: 0, 24: 1}[bits]                              # ref_driver load: 0 = int16, 1 = packed int24
        pr, info, x = run_ref_load(payload, fmt, channels, frames)
        rec = run_ref([x], cap=1)
        nei = run_ref([x], cap=1, mode="neighbours", record=1035)
        assert rec.shape[0] == nei.shape[0]
        key = str(len(names))
        names.append(name)
        out["props_" + key] = np.array([channels, rate, bits, frames], dtype=np.int64)
        out["peakrms_" + key] = pr
        out["info_" + key] = info
        out["spectral_" + key] = rec[:, 1024:].copy()
        out["neighbours_" + key] = nei[:, :11].copy()
        # rhythm front end on the analysed prefix (20 s cap does not bite for these files)
        raw = run_ref_rhythm("onsetfft", [x])
        nfr = struct.unpack("<q", raw[:8])[0]
        polar = np.frombuffer(raw[8:], dtype=np.float32).reshape(nfr, 512)
        sel = np.unique(np.concatenate([np.arange(min(24, nfr)), np.arange(0, nfr, 16), [nfr - 1]])) if nfr else np.zeros(0, np.int64)
        out["polar_frames_" + key] = sel.astype(np.int32)
        out["polar_" + key] = polar[sel]
        r = o.run_rhythm(x, original_samples=frames, data_offset=int(info[0]), cap=True)
        for t in range(2):
            series.append((f"{key}_{t}", r["sharpened"][t].copy()))
    bt = np.frombuffer(run_ref_rhythm("beattrack", [s for _, s in series]), dtype=np.float64).reshape(-1, 2)
    out["names"] = np.array(names)
    out["beat_keys"] = np.array([k for k, _ in series])
    for k, s in series:
        out["beat_in_" + k] = s
    out["beat_out"] = bt
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "real.npz"), **out)
    print("wrote tests/golden/real.npz:", {n: out["spectral_" + str(i)].shape for i, n in enumerate(names)})


if __name__ == "__main__":
    main()
