"""GPU parity of the loop's stateful neighbours: silence flag, envelope, whitening ->
peak spectrum -> spectral complexity, autocorrelation, yinfast f0 / confidence / fail-safe f0, and the
descriptors the reference's data flow makes identically zero -- through the C-ABI, against the goldens
produced by the reference's own aubio / TEnvelopeDetector / TAutocorrelation / LibXtract objects and
against the oracle on seeded batches."""
import os

import numpy as np
import pytest

import afec_amd as afx
from tests import _oracle, _tol
from tests._oracle import NEIGH_FIELDS, Oracle

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
# an impulse has an exactly flat spectrum: the whitened spectrum's local maxima are rounding noise
ILL_CONDITIONED = {("impulse", "spectral_complexity")}


@pytest.fixture(scope="module")
def plan():
    p = afx.Plan(max_analysis_ms=0)
    yield p
    p.close()


@pytest.fixture(scope="module")
def oracle():
    return Oracle()


def neighbour_names():
    z = np.load(os.path.join(GOLD, "neighbours.npz"))
    return sorted(k[4:] for k in z.files if k.startswith("ref_"))


def neighbour_input(name):
    z = np.load(os.path.join(GOLD, "neighbours.npz"))
    if "in_" + name in z.files:
        return z["in_" + name]
    return np.load(os.path.join(GOLD, "frames.npz"))["in_" + name]


def compare(res, ref, what, skip=()):
    for field, col in NEIGH_FIELDS.items():
        if field not in res or field in skip:
            continue
        rtol, atol = _tol.NEIGH_TOL[field]
        _tol.check_gpu(field, res[field], ref[:, col], rtol, atol, what=what)


@pytest.mark.parametrize("name", neighbour_names())
def test_golden_neighbours(plan, name):
    x = neighbour_input(name)            # float32, fed bit-identically to both sides
    ref = np.load(os.path.join(GOLD, "neighbours.npz"))["ref_" + name]
    res = plan.extract([x], afx.D_NEIGHBOURS)
    assert res["frame_offset"][-1] == ref.shape[0]
    compare(res, ref, name + " ", skip={f for (n, f) in ILL_CONDITIONED if n == name})


def seeded_buffers(rng):
    sr = 44100.0
    out = []
    for n, kind in [(2048 + 1024 * 7, "noise"), (5000, "tone"), (2048, "mix"), (40000, "notes"), (1500, "short"),
                    (2048 + 1024 * 30 + 333, "am"), (9000, "quiet")]:
        t = np.arange(n) / sr
        if kind == "noise":
            x = rng.uniform(-1, 1, n)
        elif kind == "tone":
            x = 0.8 * np.sin(2 * np.pi * 523.25 * t + 0.3)
        elif kind == "mix":
            x = 0.5 * np.sin(2 * np.pi * 220 * t) + 0.2 * rng.standard_normal(n)
        elif kind == "notes":
            x = np.zeros(n)
            for k, f0 in enumerate([130.8, 311.1, 87.3, 440.0]):
                a = k * 10000
                m = min(9000, n - a)
                tt = np.arange(m) / sr
                x[a:a + m] += 0.6 * np.exp(-tt / 0.08) * (np.sin(2 * np.pi * f0 * tt) + 0.5 * np.sin(4 * np.pi * f0 * tt))
        elif kind == "am":
            x = 0.7 * rng.standard_normal(n) * (0.5 + 0.5 * np.sin(2 * np.pi * 2.0 * t)) ** 4
        elif kind == "quiet":
            x = 2e-3 * rng.standard_normal(n)
        else:
            x = rng.uniform(-1, 1, n)
        out.append(x)
    return out


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_ragged_batch_against_oracle(plan, oracle, dtype):
    """several buffers of different lengths (one below a frame) in one call; every neighbour, both PCM types"""
    rng = np.random.default_rng(11)
    bufs = [b.astype(dtype) for b in seeded_buffers(rng)]
    res = plan.extract(bufs, afx.D_NEIGHBOURS)
    ref = np.concatenate([oracle.run_neighbours(b.astype(np.float64)) for b in bufs])
    assert res["frame_offset"][-1] == ref.shape[0]
    compare(res, ref, f"{np.dtype(dtype).name} ")


def test_neighbours_together_with_the_spectral_set(plan, oracle):
    """mask = everything per frame: the spectral descriptors are unchanged by the extra kernels"""
    rng = np.random.default_rng(12)
    bufs = [b.astype(np.float32) for b in seeded_buffers(rng)[:4]]
    res = plan.extract(bufs, afx.D_ALL_PER_FRAME)
    only = plan.extract(bufs, afx.D_ALL_LOW_LEVEL)
    for k, v in only.items():
        np.testing.assert_array_equal(res[k], v, err_msg=k)
    ref = np.concatenate([oracle.run_neighbours(b.astype(np.float64)) for b in bufs])
    compare(res, ref, "all ")


@pytest.mark.parametrize("bit,fields", [
    (afx.D_AMPLITUDE_SILENCE, ["amplitude_silence"]), (afx.D_AMPLITUDE_ENVELOPE, ["amplitude_envelope"]),
    (afx.D_SPECTRAL_COMPLEXITY, ["spectral_complexity"]), (afx.D_AUTO_CORRELATION, ["auto_correlation"]),
    (afx.D_F0, ["f0", "f0_confidence", "failsafe_f0"]), (afx.D_SPECTRAL_INHARMONICITY, ["spectral_inharmonicity"]),
    (afx.D_TRISTIMULUS, ["tristimulus1", "tristimulus2", "tristimulus3"])])
def test_each_neighbour_bit_alone(plan, oracle, bit, fields):
    rng = np.random.default_rng(13)
    bufs = [b.astype(np.float32) for b in seeded_buffers(rng)[:3]]
    res = plan.extract(bufs, bit)
    ref = np.concatenate([oracle.run_neighbours(b.astype(np.float64)) for b in bufs])
    assert sorted(k for k in res if k not in ("frame_offset", "buf_status")) == sorted(fields)
    compare(res, ref, "single bit ")


def test_whitening_follower_is_per_buffer_state(plan):
    """the follower restarts with every buffer (a new aubio object per file, SampleAnalyser.cpp:805-809): a
    buffer analysed alone and behind a loud one gives the same complexity series"""
    rng = np.random.default_rng(14)
    loud = rng.uniform(-1, 1, 2048 + 1024 * 12).astype(np.float32)
    soft = (0.01 * rng.standard_normal(2048 + 1024 * 12)).astype(np.float32)
    alone = plan.extract([soft], afx.D_SPECTRAL_COMPLEXITY)["spectral_complexity"]
    both = plan.extract([loud, soft], afx.D_SPECTRAL_COMPLEXITY)
    np.testing.assert_array_equal(both["spectral_complexity"][both["frame_offset"][1]:], alone)


def test_autocorrelation_looks_past_the_last_frame_when_the_buffer_has_more(plan, oracle):
    """CalcAutoCorrelation's second rising-slope search may read up to 33 samples past the frame
    (remaining = size - n, SampleAnalyser.cpp:943): a falling ramp after a late first rise forces it"""
    n = 2048 + 1024 + 700                       # 2 frames, 700 samples beyond the last one
    x = np.linspace(1.0, -1.0, n)               # falling everywhere, except for two upward steps:
    x[2025:] += 0.01                            # sample 1000 of the last frame (its first rise), and
    x[3077:] += 0.01                            # 5 samples past that frame's end (found by the second search)
    res = plan.extract([x.astype(np.float64)], afx.D_AUTO_CORRELATION)
    ref = oracle.run_neighbours(x)
    _tol.check("auto_correlation", res["auto_correlation"], ref[:, NEIGH_FIELDS["auto_correlation"]], 1e-6, 1e-9)
    # and the same frames without the tail give another answer for the last frame (the step is not seen)
    cut = oracle.run_neighbours(x[:3072])
    assert abs(cut[1, NEIGH_FIELDS["auto_correlation"]] - ref[1, NEIGH_FIELDS["auto_correlation"]]) > 1e-6


def test_statistics_of_the_neighbour_series(plan, oracle):
    rng = np.random.default_rng(15)
    bufs = [b.astype(np.float32) for b in seeded_buffers(rng)[:4]]
    mask = afx.D_NEIGHBOURS | afx.D_STATISTICS
    b = plan.batch(bufs, mask)
    b.run()
    res, st = b.fetch(), b.fetch_statistics()
    b.close()
    off = res["frame_offset"]
    for i in range(len(bufs)):
        for field in NEIGH_FIELDS:
            series = res[field][off[i]:off[i + 1]]
            want = _oracle.calc_statistics(series)
            got = st[field][i]
            if off[i + 1] - off[i] >= 2:
                # the GPU series itself is the input of both sides: only the reduction is compared
                skip_ill = abs(series.sum()) < 1e-6 * np.abs(series).sum() if series.size else True
                for j, name in enumerate(afx.STAT_NAMES):
                    if skip_ill and name in ("centroid", "spread", "skewness", "kurtosis", "flatness"):
                        continue
                    assert abs(got[j] - want[j]) <= 1e-9 * abs(want[j]) + 1e-12, (i, field, name, got[j], want[j])


def test_from_raw_front_end_feeds_the_neighbours(plan, oracle):
    rng = np.random.default_rng(16)
    t = np.arange(30000) / 44100.0
    pcm = (0.4 * np.sin(2 * np.pi * 330 * t) * np.exp(-t / 0.2) + 0.01 * rng.standard_normal(t.size))
    raw = np.round(pcm * 20000).astype(np.int16)
    b, infos = plan.batch_from_raw([(raw, 1)], afx.D_NEIGHBOURS)
    b.run()
    res = b.fetch()
    mono, info = _oracle.load_sample(raw, 1)
    b.close()
    ref = oracle.run_neighbours(mono)
    assert res["frame_offset"][-1] == ref.shape[0]
    compare(res, ref, "from_raw ")


def test_empty_and_short_inputs(plan):
    res = plan.extract([], afx.D_NEIGHBOURS)
    assert res["f0"].shape == (0,) and res["frame_offset"].tolist() == [0]
    res = plan.extract([np.zeros(100, np.float32), np.zeros(2047, np.float32)], afx.D_ALL_PER_FRAME)
    assert res["spectral_complexity"].shape == (0,) and res["frame_offset"].tolist() == [0, 0, 0]


def test_exactly_one_frame_and_the_20s_cap(oracle):
    """remaining = size - n reaches its minimum (2048) on a one-frame buffer; with the cap the buffer is longer
    than its analysed prefix and CalcAutoCorrelation still sees the real size (SampleAnalyser.cpp:943)"""
    rng = np.random.default_rng(21)
    one = rng.uniform(-1, 1, 2048)
    capped = afx.Plan(max_analysis_ms=20000)
    res = capped.extract([one], afx.D_NEIGHBOURS)
    compare(res, oracle.run_neighbours(one, cap=True), "one frame ")
    long = (0.3 * rng.standard_normal(882000 + 5000)).astype(np.float32)      # 20 s + a bit
    res = capped.extract([long], afx.D_AUTO_CORRELATION | afx.D_AMPLITUDE_SILENCE)
    ref = oracle.run_neighbours(long.astype(np.float64), cap=True)
    assert res["frame_offset"][-1] == ref.shape[0] == 860
    compare(res, ref, "capped ")
    capped.close()


def test_the_float_stft_mode_is_gone():
    """AFX_PRECISION_F32 (round 1) missed the parity bar on tonal input and is no faster than the double half-wave
    kernel: plans asking for it are refused, loudly."""
    with pytest.raises(afx.AfxError) as ei:
        afx.Plan(max_analysis_ms=0, precision=afx.PRECISION_F32)
    assert ei.value.status == -2

def test_non_finite_samples_do_not_poison_other_frames_or_buffers(plan, oracle):
    """a NaN / Inf sample is garbage in the frames that contain it (as in the reference) and nowhere else"""
    rng = np.random.default_rng(23)
    good = rng.uniform(-1, 1, 2048 + 1024 * 6)
    bad = good.copy()
    bad[5000] = np.nan          # frames 3 and 4 contain sample 5000
    bad[5001] = np.inf
    res = plan.extract([bad, good], afx.D_NEIGHBOURS)
    ref = oracle.run_neighbours(good)
    off = res["frame_offset"]
    clean = [0, 1, 2, 5, 6]
    for field, col in NEIGH_FIELDS.items():
        if field == "spectral_complexity":
            continue            # the follower carries the poisoned frames forward, as the reference's would
        rtol, atol = _tol.NEIGH_TOL[field]
        _tol.check_gpu(field, res[field][off[0]:off[1]][clean], ref[clean, col], rtol, atol, what="frames without the NaN ")
        _tol.check_gpu(field, res[field][off[1]:off[2]], ref[:, col], rtol, atol, what="second buffer ")
    assert np.all(np.isfinite(res["spectral_complexity"]))


def test_effective_length_golden_and_beyond_the_cap(oracle):
    """per-file effective lengths (AFX_D_EFFECTIVE_LENGTH) against the reference goldens; the scan covers the
    whole buffer, also the part beyond the analysed 20 s"""
    z = np.load(os.path.join(GOLD, "efflen.npz"))
    names = sorted(k[4:] for k in z.files if k.startswith("ref_"))
    inputs = [z["in_" + n] if "in_" + n in z.files else neighbour_input(n) for n in names]
    capped = afx.Plan(max_analysis_ms=20000)
    res = capped.extract(inputs, afx.D_EFFECTIVE_LENGTH | afx.D_MFCC)
    np.testing.assert_array_equal(res["effective_length"], np.stack([z["ref_" + n] for n in names]))
    alone = capped.extract(inputs, afx.D_EFFECTIVE_LENGTH)      # no per-frame descriptor at all
    np.testing.assert_array_equal(alone["effective_length"], res["effective_length"])
    rng = np.random.default_rng(31)
    long = np.zeros(882000 + 60000)
    long[1000:882000 + 50000] = 0.2 * rng.standard_normal(882000 + 49000)     # audible well past 20 s
    short = np.zeros(700); short[100:300] = 0.5                                # shorter than a frame
    for dt in (np.float32, np.float64):
        bufs = [long.astype(dt), short.astype(dt), np.zeros(3000, dt)]
        got = capped.extract(bufs, afx.D_EFFECTIVE_LENGTH | afx.D_SPECTRAL_RMS)["effective_length"]
        want = np.stack([oracle.effective_length(b.astype(np.float64)) for b in bufs])
        np.testing.assert_array_equal(got, want)
        assert want[0, 0] > 20.5 and np.all(want[2] == 0.0)
    capped.close()


def test_effective_length_of_many_buffers(oracle):
    """batches of 64 and more buffers take the kernel that searches from both ends (afx_load.hip): decays that end above
    one floor and below the next, floors that are never crossed, silence, one-sample bursts at either end, lengths around
    the 1024-sample blocks"""
    rng = np.random.default_rng(57)
    bufs = []
    for i in range(90):
        n = int(rng.choice([1, 5, 1023, 1024, 1025, 4096, 30000, 70001]))
        kind = i % 6
        if kind == 0:
            x = rng.uniform(-1, 1, n) * np.exp(-np.arange(n) / max(1.0, n / rng.uniform(2, 30)))
        elif kind == 1:
            x = rng.uniform(-1, 1, n) * rng.choice([0.003, 0.02, 0.1])          # below some or all of the floors
        elif kind == 2:
            x = np.zeros(n)
        elif kind == 3:
            x = np.zeros(n); x[0] = 0.9; x[-1] = rng.choice([0.9, 0.05, 0.005])
        elif kind == 4:
            x = np.zeros(n); x[n // 2] = 0.3                                       # one sample in the middle
        else:
            x = np.concatenate([np.zeros(n // 3), rng.uniform(-1, 1, n - n // 3) * np.linspace(0, 1, n - n // 3) ** 4])
        bufs.append(x)
    p = afx.Plan(max_analysis_ms=0)
    for dt in (np.float32, np.float64):
        typed = [b.astype(dt) for b in bufs]
        got = p.extract(typed, afx.D_EFFECTIVE_LENGTH)["effective_length"]
        want = np.stack([oracle.effective_length(b.astype(np.float64)) for b in typed])
        np.testing.assert_array_equal(got, want)
        few = p.extract(typed[:40], afx.D_EFFECTIVE_LENGTH)["effective_length"]     # the kernel for few buffers
        np.testing.assert_array_equal(few, want[:40])
    p.close()


def test_chunking_invariance_bitwise(plan):
    """the same buffer alone (short chunks, one round) and inside a large batch (long chunks): carried state --
    the whitening follower, the pitch kernel's first-half transform -- must not depend on where chunks start"""
    rng = np.random.default_rng(41)
    x = (0.4 * rng.standard_normal(2048 + 1024 * 2999)).astype(np.float32)
    x[200000:260000] *= 0.001                       # a quiet stretch: the follower decays through it
    alone = plan.extract([x], afx.D_NEIGHBOURS)
    others = [rng.uniform(-1, 1, 2048 + 1024 * 1499).astype(np.float32) for _ in range(40)]
    batch = plan.extract(others[:20] + [x] + others[20:], afx.D_NEIGHBOURS)
    lo, hi = batch["frame_offset"][20], batch["frame_offset"][21]
    assert hi - lo == 3000
    for k in NEIGH_FIELDS:
        np.testing.assert_array_equal(batch[k][lo:hi], alone[k], err_msg=k)


def test_full_size_buffer_against_oracle_samples(plan, oracle):
    """BASELINE configs[1] size (one 10 000-frame buffer): the head of the buffer against the oracle for every
    neighbour, and frames deep inside it for the descriptors that only see their own frame"""
    rng = np.random.default_rng(42)
    n = 2048 + 1024 * 9999
    t = np.arange(n) / 44100.0
    x = (0.3 * np.sin(2 * np.pi * (180.0 + 40.0 * np.sin(2 * np.pi * 0.7 * t)) * t) + 0.05 * rng.standard_normal(n)).astype(np.float32)
    res = plan.extract([x], afx.D_NEIGHBOURS)
    assert res["frame_offset"][-1] == 10000
    head = oracle.run_neighbours(x[:2048 + 1024 * 63].astype(np.float64))
    for field, col in NEIGH_FIELDS.items():
        if field == "auto_correlation":
            continue        # `remaining` differs between the slice and the whole buffer only in the last frame; checked below
        rtol, atol = _tol.NEIGH_TOL[field]
        _tol.check_gpu(field, res[field][:64], head[:, col], rtol, atol, what="head ")
    local = ["amplitude_silence", "amplitude_envelope", "f0", "f0_confidence", "auto_correlation"]
    for f in rng.integers(100, 9990, 24):
        seg = x[1024 * f: 1024 * f + 2048 + 64].astype(np.float64)
        ref = oracle.run_neighbours(seg)[0]
        for field in local:
            rtol, atol = _tol.NEIGH_TOL[field]
            _tol.check_gpu(field, res[field][f:f + 1], ref[NEIGH_FIELDS[field]:NEIGH_FIELDS[field] + 1], rtol, atol,
                       what=f"frame {f} ")
