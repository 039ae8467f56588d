"""SURVEY 8(f) row f4, last part: the rhythm tracker (SampleAnalyser.cpp:983-1048) on the GPU (afx_rhythm.hip),
through the C-ABI (AFX_D_RHYTHM, afx_batch_fetch_rhythm), against the oracle (oracle/afx_oracle_rhythm.c).

Tolerances: the onset functions are float in the reference.  The power function is float multiply / add in a fixed
order and must agree bit for bit wherever the FFT front ends round the same; the complex-domain function contains cosf
(the GPU rounds a double cosine, 1 ulp off glibc's cosf for ~1.3 % of arguments) and agrees to a few float ulps.
Detected onset positions must be identical; everything downstream is double arithmetic on those series and agrees far
inside the north-star bar of 1e-4 relative."""
import os

import numpy as np
import pytest

import afec_amd as afx
from tests import _oracle

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "rhythm.npz"))
NAMES = ["loop120", "loop95", "oneshot", "melody"]
MASK = afx.D_RHYTHM | afx.D_STATISTICS


def signal(name):
    return GOLD[f"pcm_{name}"].astype(np.float64) / 32768.0


def check_file(name, x, onsets, odf, scalars, ref):
    T = ref["onsets"].shape[1]
    assert onsets.shape == (T, 2), name
    # onset functions before median removal
    o_ref = ref["odf"].T.astype(np.float32)
    scale = np.abs(o_ref).max(axis=0) + 1e-30
    assert np.all(np.abs(odf.astype(np.float64) - o_ref) <= 2e-6 * np.abs(o_ref) + 2e-7 * scale), name
    assert np.mean(odf[:, 1] == o_ref[:, 1]) > 0.98, (name, "power function should be bit-equal almost everywhere")
    # detections: same frames, values to float rounding of (value - median)
    for t in range(2):
        assert np.array_equal(np.nonzero(onsets[:, t])[0], np.nonzero(ref["onsets"][t])[0]), (name, t)
        assert np.all(np.abs(onsets[:, t] - ref["onsets"][t]) <= 4e-6 * scale[t]), (name, t)
    want = ref["scalars"]
    for k, (g, w) in enumerate(zip(scalars, want)):
        label = afx.capi.RHYTHM_SCALARS[k]
        if "onset_count" in label:
            assert g == w, (name, label, g, w)
        else:
            assert abs(g - w) <= 1e-5 * abs(w) + 1e-9, (name, label, g, w)


def test_rhythm_against_oracle_f64_batch():
    plan = afx.Plan()
    xs = [signal(n) for n in NAMES]
    b = plan.batch(xs, MASK)
    b.run()
    r = b.fetch_rhythm(statistics=True, onset_functions=True)
    o = _oracle.Oracle()
    off = r["offsets"]
    assert off.tolist() == np.concatenate([[0], np.cumsum([o.rhythm_frames(x.size, cap=True) for x in xs])]).tolist()
    for i, (name, x) in enumerate(zip(NAMES, xs)):
        ref = o.run_rhythm(x, cap=True)
        sl = slice(off[i], off[i + 1])
        check_file(name, x, r["onsets"][sl], r["onset_functions"][sl], r["scalars"][i], ref)
        # CalcStatistics covers the two onset series (SampleAnalyser.cpp:2402-2411)
        for t in range(2):
            want = _oracle.calc_statistics(r["onsets"][sl, t], np.zeros(13))
            assert np.all(np.abs(r["onset_statistics"][i, t] - want) <= 1e-9 * np.abs(want) + 1e-12), (name, t)
    b.close()
    plan.close()


def test_rhythm_f32_pcm_and_repeated_runs():
    plan = afx.Plan()
    xs = [signal(n).astype(np.float32) for n in ("loop95", "loop120")]
    b = plan.batch(xs, afx.D_RHYTHM | afx.D_MFCC)
    b.run()
    first = b.fetch_rhythm(onset_functions=True)
    b.run()
    second = b.fetch_rhythm(onset_functions=True)
    for k in ("onsets", "scalars", "onset_functions"):
        assert np.array_equal(first[k], second[k]), k
    o = _oracle.Oracle()
    off = first["offsets"]
    for i, x in enumerate(xs):
        ref = o.run_rhythm(x.astype(np.float64), cap=True)
        sl = slice(off[i], off[i + 1])
        check_file(f"f32[{i}]", x, first["onsets"][sl], first["onset_functions"][sl], first["scalars"][i], ref)
    b.close()
    plan.close()


def test_rhythm_file_info_drives_the_final_tempo():
    plan = afx.Plan()
    x = signal("loop120")
    o = _oracle.Oracle()
    b = plan.batch([x, x, x], afx.D_RHYTHM)
    b.set_file_info([(44100, 0, x.size), (44100, 0, 2 * x.size), (22050, -3000, x.size)])
    b.run()
    r = b.fetch_rhythm()
    refs = [o.run_rhythm(x, cap=True), o.run_rhythm(x, original_samples=2 * x.size, cap=True),
            o.run_rhythm(x, original_rate=22050, data_offset=-3000, cap=True)]
    for i, ref in enumerate(refs):
        for g, w in zip(r["scalars"][i], ref["scalars"]):
            assert abs(g - w) <= 1e-5 * abs(w) + 1e-9, (i, r["scalars"][i], ref["scalars"])
    assert abs(r["scalars"][0][12] - 120.0) < 0.01
    b.close()
    plan.close()


def test_rhythm_from_raw_uses_the_load_information():
    """LoadSample on the GPU, then the rhythm tracker: duration and onset offset come from the file (SampleAnalyser.cpp:
    1001-1004), and the analysed buffer is the normalised, trimmed, padded one."""
    plan = afx.Plan()
    pcm = np.concatenate([np.zeros(3000, dtype=np.int16), GOLD["pcm_loop95"]])
    b, infos = plan.batch_from_raw([(pcm, 1)], afx.D_RHYTHM)
    b.run()
    r = b.fetch_rhythm(onset_functions=True)
    x, info = _oracle.load_sample(pcm, 1)
    assert info["data_offset"] == infos[0]["data_offset"] and info["data_offset"] < 0
    got_x = b.fetch_samples(0, x.size)
    assert np.array_equal(got_x, x)
    ref = _oracle.Oracle().run_rhythm(x, original_samples=pcm.size, data_offset=info["data_offset"], cap=True)
    check_file("raw", x, r["onsets"], r["onset_functions"], r["scalars"][0], ref)
    b.close()
    plan.close()


def test_rhythm_edge_cases():
    plan = afx.Plan()
    rng = np.random.default_rng(5)
    bufs = [np.zeros(0), np.zeros(300), np.zeros(512), np.zeros(44100), 0.5 * rng.uniform(-1, 1, 511 + 128 * 3),
            1e-6 * rng.uniform(-1, 1, 30000)]
    b = plan.batch(bufs, MASK)
    b.run()
    r = b.fetch_rhythm(statistics=True, onset_functions=True)
    o = _oracle.Oracle()
    assert np.diff(r["offsets"]).tolist() == [0, 0, 1, (44100 - 512) // 128 + 1, 3, (30000 - 512) // 128 + 1]
    assert np.all(np.isfinite(r["scalars"]))
    for i, x in enumerate(bufs):
        if x.size < 512:
            assert np.all(r["scalars"][i] == 0.0)
            continue
        ref = o.run_rhythm(x, cap=True)
        sl = slice(r["offsets"][i], r["offsets"][i + 1])
        check_file(f"edge[{i}]", x, r["onsets"][sl], r["onset_functions"][sl], r["scalars"][i], ref)
    b.close()
    plan.close()


def test_rhythm_long_file_hits_the_cap():
    # 25 s: the analysed prefix is 20 s = 6887 frames; the autocorrelation of the beat tracker runs over all of them
    sr = 44100
    rng = np.random.default_rng(9)
    n = 25 * sr
    x = np.zeros(n)
    step = int(60.0 / 132.0 * sr)
    for at in range(0, n - 4000, step):
        x[at:at + 2500] += np.exp(-np.arange(2500) / 300.0) * rng.uniform(-1, 1, 2500)
    x /= np.abs(x).max()
    plan = afx.Plan()
    b = plan.batch([x.astype(np.float32)], MASK)
    b.run()
    r = b.fetch_rhythm(onset_functions=True)
    ref = _oracle.Oracle().run_rhythm(x.astype(np.float32).astype(np.float64), cap=True)
    assert r["offsets"][-1] == (882000 - 512) // 128 + 1
    check_file("long", x, r["onsets"], r["onset_functions"], r["scalars"][0], ref)
    assert abs(r["scalars"][0][7] - 132.0) < 2.0
    b.close()
    plan.close()


def test_rhythm_without_the_cap_series_longer_than_the_lds_stage():
    # 75 s with the 20 s cap disabled: 25 836 onset frames per function -- the autocorrelation reads the series from
    # global memory (it no longer fits the 8 192-frame LDS stage) and the onset statistics take the long-series kernel
    sr = 44100
    rng = np.random.default_rng(19)
    n = 75 * sr
    x = np.zeros(n)
    step = int(60.0 / 104.0 * sr)
    for k, at in enumerate(range(0, n - 5000, step // 2)):
        amp = 1.0 if k % 2 == 0 else 0.35
        x[at:at + 3000] += amp * np.exp(-np.arange(3000) / 350.0) * rng.uniform(-1, 1, 3000)
    x = (x / np.abs(x).max()).astype(np.float32)
    plan = afx.Plan(max_analysis_ms=0)
    b = plan.batch([x, x[:200000]], MASK)
    b.run()
    r = b.fetch_rhythm(statistics=True, onset_functions=True)
    o = _oracle.Oracle()
    assert np.diff(r["offsets"]).tolist() == [(n - 512) // 128 + 1, (200000 - 512) // 128 + 1]
    for i, sig in enumerate((x, x[:200000])):
        ref = o.run_rhythm(sig.astype(np.float64), cap=False)
        sl = slice(r["offsets"][i], r["offsets"][i + 1])
        check_file(f"uncapped[{i}]", sig, r["onsets"][sl], r["onset_functions"][sl], r["scalars"][i], ref)
        for t in range(2):
            want = _oracle.calc_statistics(r["onsets"][sl, t], np.zeros(13))
            assert np.all(np.abs(r["onset_statistics"][i, t] - want) <= 1e-8 * np.abs(want) + 1e-12), (i, t)
    b.close()
    plan.close()
