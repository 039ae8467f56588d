"""CPU tests of the rhythm-tracker oracle (oracle/afx_oracle_rhythm.c; SampleAnalyser.cpp:983-1048).

Pinned parts against reference-generated goldens (tests/golden/rhythm.npz, made by make_golden_rhythm.py from the
reference's own window / ooura_cdft / TAudioMath / aubio beattracking objects); the parts whose reference classes do not
link here (TOnsetDetector, TCannyWindow, TRhythmTracker heuristics: "parity unpinned") against an independent numpy
restatement in float32 and against hand-computed cases."""
import os

import numpy as np
import pytest

from . import _oracle

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "rhythm.npz"))
NAMES = ["loop120", "loop95", "oneshot", "melody"]


def signal(name):
    return GOLD[f"pcm_{name}"].astype(np.float64) / 32768.0


@pytest.mark.parametrize("name", NAMES)
def test_onset_polar_matches_reference_objects(name):
    x = signal(name)
    frames, ref = GOLD[f"polar_frames_{name}"], GOLD[f"polar_{name}"]
    got = np.stack([_oracle.onset_polar(x[f * 128:f * 128 + 512]) for f in frames])
    # dc, "nyquist" (= Im[0], always 0) and the 255 magnitudes: equal floats except where the two FFT algorithms
    # round the last bit of the double differently
    mag_ref, mag_got = ref[:, :257].astype(np.float64), got[:, :257].astype(np.float64)
    scale = np.abs(mag_ref[:, 2:]).max(axis=1, keepdims=True) + 1e-30
    assert np.mean(ref[:, :257] == got[:, :257]) > 0.995
    assert np.all(np.abs(mag_ref - mag_got) <= 1.2e-7 * np.abs(mag_ref) + 1e-13 * scale)
    assert np.all(ref[:, 1] == 0.0) and np.all(got[:, 1] == 0.0)
    # phases: only meaningful where the bin is above the rounding noise of the frame
    loud = mag_ref[:, 2:] > 1e-9 * scale
    d = np.abs(ref[:, 257:].astype(np.float64) - got[:, 257:].astype(np.float64))
    d = np.minimum(d, 2 * np.pi - d)
    assert np.all(d[loud] <= 5e-7)
    assert np.mean((ref[:, 257:] == got[:, 257:])[loud]) > 0.995


def test_beat_tracking_matches_reference_aubio():
    for name, ref in zip(GOLD["beat_names"], GOLD["beat_out"]):
        bpm, conf = _oracle.beattrack(GOLD[f"beat_in_{name}"])
        assert bpm == pytest.approx(ref[0], rel=1e-13, abs=0), name
        assert conf == pytest.approx(ref[1], rel=1e-12, abs=1e-300), name


def test_rhythm_frame_count():
    o = _oracle.Oracle()
    assert o.rhythm_frames(511) == 0
    assert o.rhythm_frames(512) == 1
    assert o.rhythm_frames(512 + 127) == 1
    assert o.rhythm_frames(512 + 128) == 2
    # 20 s cap of the analysed prefix (SampleAnalyser.cpp:760-764)
    assert o.rhythm_frames(44100 * 30, cap=True) == (882000 - 512) // 128 + 1


def _numpy_detector(x, rate=44100):
    """Independent float32 restatement of Whiten + kFunctionPower + DetectOnset (OnsetDetector.cpp:186-240, 380-388,
    549-587), vectorised over bins / frames; the polar front end comes from the (pinned) oracle function."""
    f32 = np.float32
    T = (x.size - 512) // 128 + 1
    polar = np.stack([_oracle.onset_polar(x[f * 128:f * 128 + 512]) for f in range(T)])
    dc, mag = polar[:, 0], polar[:, 2:257]
    coef = f32(np.exp((np.float64(-2.30258509) * np.float64(f32(128))) / np.float64(f32(25.0) * f32(rate))))
    psp = np.zeros(257)
    odf = np.zeros(T, dtype=f32)
    for f in range(T):
        cur = np.concatenate([[abs(dc[f])], np.abs(mag[f]), [0.0]]).astype(np.float64)   # psp[0], psp[1..255], psp[256]
        dec = cur + (psp - cur) * np.float64(coef)
        psp = np.where(cur < psp, dec, cur)
        div = np.maximum(np.float64(f32(0.1)), psp).astype(f32)
        wdc = f32(dc[f]) / div[0]
        wm = mag[f] / div[1:256]
        acc = f32(0.0) * f32(0.0) + wdc * wdc
        for i in range(255):
            acc = f32(acc + wm[i] * wm[i])
        odf[f] = acc * (f32(2560.0) / f32(257 * 512))
    med = max(3, int((f32(rate) * f32(0.2)) / f32(128) + f32(0.5)))
    gap_len = int((f32(rate) * f32(0.12)) / f32(128) + f32(0.5))
    padded = np.concatenate([np.zeros(med - 1, dtype=f32), odf])
    onsets = np.zeros(T)
    prev, gap = f32(0.0), 0
    for f in range(T):
        w = np.sort(padded[f:f + med])
        m = w[(med - 1) >> 1] if med & 1 else f32((w[med >> 1] + w[(med >> 1) - 1]) * f32(0.5))
        post = f32(odf[f] - m)
        det = False
        if gap:
            gap -= 1
        elif post > f32(0.8) and prev <= f32(0.8):
            det, gap = True, gap_len
        onsets[f] = float(post) if det else 0.0
        prev = post
    return odf, onsets


@pytest.mark.parametrize("name", ["loop120", "oneshot"])
def test_power_detector_against_numpy_float32(name):
    x = signal(name)
    r = _oracle.Oracle().run_rhythm(x)
    odf, onsets = _numpy_detector(x)
    assert np.array_equal(r["odf"][1].astype(np.float32), odf)
    assert np.array_equal(r["onsets"][1], onsets)
    assert r["scalars"][6] == np.count_nonzero(onsets > 0.8)


def test_canny_window_by_hand():
    # impulse response: tmp[i] = w[p - i + 12] for -12 <= p - i < 12, then (x - mean) / std clipped at 0
    n, p = 64, 30
    x = np.zeros(n)
    x[p] = 2.0
    k = np.arange(-12, 13)
    w = k / 256.0 * np.exp(-1.0 * (k * k) / (2.0 * 256.0))
    tmp = np.zeros(n)
    for i in range(n):
        s = p - i
        if -12 <= s < 12:
            tmp[i] = 2.0 * w[s + 12]
    z = (tmp - tmp.mean()) / np.sqrt(((tmp - tmp.mean()) ** 2).mean())
    np.testing.assert_allclose(_oracle.canny(x), np.maximum(0.0, z), rtol=1e-13, atol=1e-15)
    # a constant series convolves to something non-constant at the borders only; an all-zero one stays zero
    assert np.all(_oracle.canny(np.zeros(40)) == 0.0)


def test_click_track_tempo_and_counts():
    o = _oracle.Oracle()
    for name, bpm in [("loop120", 120.0), ("loop95", 95.0)]:
        x = signal(name)
        r = o.run_rhythm(x)
        s = dict(zip(_oracle.RHYTHM_SCALARS, r["scalars"]))
        beats = x.size / 44100.0 * bpm / 60.0
        assert 0.5 * beats <= s["rhythm_percussive_onset_count"] <= 2.2 * beats
        assert abs(s["rhythm_final_tempo"] - bpm) < 1.5, (name, s)
        assert s["rhythm_final_tempo_confidence"] >= 0.5
        # minimum gap between detections: 21 / 41 frames (OnsetDetector.cpp:272)
        for t, gap in ((0, 21), (1, 41)):
            pos = np.nonzero(r["onsets"][t])[0]
            assert np.all(np.diff(pos) > gap)
        assert -1.0 < s["rhythm_percussive_onset_contrast"] < 0.0


def test_silence_and_one_shot():
    o = _oracle.Oracle()
    r = o.run_rhythm(np.zeros(44100))
    assert np.all(r["onsets"] == 0.0) and np.all(r["scalars"] == 0.0)
    r = o.run_rhythm(signal("oneshot"))
    s = dict(zip(_oracle.RHYTHM_SCALARS, r["scalars"]))
    assert s["rhythm_complex_onset_count"] <= 2 and s["rhythm_percussive_onset_count"] <= 2
    assert s["rhythm_final_tempo"] == 0.0 and s["rhythm_final_tempo_confidence"] == 0.0   # fewer than 4 onsets


def test_duration_heuristics_depend_on_file_info():
    # the same samples described as a file of twice the duration: the guessed number of beats changes the final tempo
    o = _oracle.Oracle()
    x = signal("loop120")
    a = o.run_rhythm(x)["scalars"]
    b = o.run_rhythm(x, original_samples=2 * x.size)["scalars"]
    assert np.array_equal(a[:12], b[:12])          # only the final tempo sees the file information
    assert abs(a[12] - 120.0) < 0.01               # 8 beats in 4.0 s
    assert b[12] != a[12]
    # a data offset moves the expected onset grid (RhythmTracker.cpp:490-503)
    c = o.run_rhythm(x, data_offset=-5000)["scalars"]
    assert np.array_equal(a[:12], c[:12])
