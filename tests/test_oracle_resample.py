"""The oracle's restatement of the sample-rate conversion in LoadSample (oracle/afx_oracle_resample.c; SampleAnalyser.cpp:
563-607 -> libresample 0.1.3) against the reference's own libresample: tests/golden/resample.npz was written by
tests/golden/make_golden_resample.py from oracle/_ref/ref_driver.  Bit-exact.  CPU only."""
import hashlib
import os

import numpy as np
import pytest

from tests import _oracle
from tests.golden.make_golden_resample import signal

GOLD = os.path.join(os.path.dirname(__file__), "golden", "resample.npz")


def golden_cases():
    z = np.load(GOLD)
    return z, [tuple(int(v) for v in row) for row in z["cases"]]


def check_against_golden(z, case, y, written):
    rate, n, _, new_size, want_written, _ = case
    key = f"{rate}_{n}"
    assert y.size == new_size and written == want_written, key
    if "out_" + key in z:
        np.testing.assert_array_equal(y.view(np.uint32), z["out_" + key].view(np.uint32), err_msg=key)
    else:
        np.testing.assert_array_equal(y[:64].view(np.uint32), z["head_" + key].view(np.uint32), err_msg=key)
        np.testing.assert_array_equal(y[-64:].view(np.uint32), z["tail_" + key].view(np.uint32), err_msg=key)
        assert hashlib.sha256(y.tobytes()).digest() == z["sha_" + key].tobytes(), key


def test_oracle_equals_the_references_libresample():
    z, cases = golden_cases()
    assert len(cases) >= 120
    for case in cases:
        rate, n, seed = case[:3]
        y, written = _oracle.resample(signal(n, seed), rate)
        check_against_golden(z, case, y, written)


def test_new_size_and_the_identity_rate():
    """NewSizeInSamples = max(1, d2iRound(n / Speed)) (SA:572-573); a file at the analyser's rate is not converted."""
    for rate, n in [(48000, 1), (96000, 1), (192000, 2), (22050, 3), (48000, 48000), (8000, 8000)]:
        y, _ = _oracle.resample(np.ones(n, np.float32), rate)
        want = max(1, int(n / (rate / 44100.0) + 0.5))
        assert y.size == want, (rate, n)
    x = np.round(np.random.default_rng(3).uniform(-20000, 20000, 5000)).astype(np.int16)
    a, ia = _oracle.load_sample(x, 1)
    b, ib = _oracle.load_sample(x, 1, file_rate=44100, rate=44100)
    np.testing.assert_array_equal(a, b)
    assert ia == ib


def test_dc_gain_and_band_limit():
    """What the conversion is for: unity gain in the pass band, nothing above the new Nyquist frequency."""
    for rate in (48000, 96000, 22050):
        n = rate // 2
        y, _ = _oracle.resample(np.full(n, 1000.0, np.float32), rate)
        mid = y[y.size // 4: 3 * y.size // 4]
        assert np.max(np.abs(mid - 1000.0)) < 2.0, rate                # Kaiser-windowed sinc: ~ -60 dB pass-band ripple
        t = np.arange(n) / rate
        tone = (10000.0 * np.sin(2 * np.pi * 1000.0 * t)).astype(np.float32)
        y, _ = _oracle.resample(tone, rate)
        want = 10000.0 * np.sin(2 * np.pi * 1000.0 * np.arange(y.size) / 44100.0)
        assert np.max(np.abs(y[200:-200] - want[200:-200])) < 30.0, rate
    t = np.arange(48000) / 96000.0
    high = (10000.0 * np.sin(2 * np.pi * 30000.0 * t)).astype(np.float32)   # above 22.05 kHz: must not alias
    y, _ = _oracle.resample(high, 96000)
    assert np.max(np.abs(y[200:-200])) < 100.0


def test_filter_table():
    imp = _oracle.resample_filter()
    assert imp.size == 4096 * 17 and abs(float(imp[0]) - 0.9) < 1e-7          # c[0] = 2 frq = 0.9 (filterkit.c:88)
    assert np.all(np.abs(imp) <= imp[0]) and abs(imp[-1]) < 1e-3


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "ref_driver")),
                    reason="needs oracle/_ref/ref_driver (make -C oracle ref; only in the build container)")
def test_oracle_equals_the_references_libresample_on_random_cases():
    """Beyond the committed goldens: random rates and lengths through the reference's libresample on the spot."""
    from tests.golden.make_golden_resample import run_ref_resample
    rng = np.random.default_rng(2026)
    for _ in range(60):
        rate = int(rng.choice([8000, 11025, 16000, 22050, 32000, 48000, 88200, 96000, 176400, 192000])) if rng.random() < 0.5 else int(rng.integers(3000, 400000))
        n = int(rng.choice([1, 2, 3, 4039, 4040, 4041])) if rng.random() < 0.2 else int(np.exp(rng.uniform(np.log(5), np.log(60000))))
        x = np.round(rng.uniform(-30000, 30000, n)).astype(np.float32)
        want, written, used = run_ref_resample(x, rate)
        got, got_written = _oracle.resample(x, rate)
        assert got.size == want.size and got_written == written and used == n, (rate, n)
        np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32), err_msg=f"{rate} Hz, {n} samples")
