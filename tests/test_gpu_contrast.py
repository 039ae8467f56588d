"""Sub-band contrast (SampleAnalyser.cpp:2200-2232): the mean of the lowest / highest 30 % of a band's bins is an
exact selection on the original doubles (afx_bands.hip: the 27-bit sort keys only locate the cut).  The contrast
raises peak / valley to 1 / ln(band mean), so the test includes bands whose mean sits next to 1.0, and inputs
whose bins tie in the sort keys (silence: all zero; an impulse: a flat spectrum up to rounding)."""
import os

import numpy as np
import pytest

import afec_amd as afx
from tests import _tol
from tests._oracle import FIELDS, Oracle

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
MASK = afx.D_BAND_FEATURES | afx.D_MAGNITUDE


@pytest.fixture(scope="module")
def plan():
    p = afx.Plan(max_analysis_ms=0)
    yield p
    p.close()


def contrast_from_magnitudes(mag):
    """The reference's arithmetic (SA:2200-2232) on a given magnitude spectrum, in numpy: isolates the selection
    and the sums from the FFT's rounding."""
    counts = [2, 4, 6, 10, 12, 15, 17, 23, 29, 41, 61, 96, 148, 287]
    out = np.zeros((mag.shape[0], 14))
    for f in range(mag.shape[0]):
        k = 1
        for b, n in enumerate(counts):
            band = np.sort(mag[f, k:k + n])
            nn = max(1, int(0.3 * n))
            valley = float(np.sum(band[:nn])) / nn + 1e-30
            peak = float(np.sum(band[::-1][:nn])) / nn + 1e-30
            mean = float(np.sum(mag[f, k:k + n])) / n if n >= 2 else float(mag[f, k])
            with np.errstate(all="ignore"):
                out[f, b] = -1.0 * np.power(peak / valley, 1.0 / np.log(mean + 1e-30))
            k += n
    return out


def check_against_own_magnitudes(plan, x, what, rtol=1e-11):
    res = plan.extract([x], MASK)
    want = contrast_from_magnitudes(res["magnitude"])
    got = res["sub_contrast"].reshape(want.shape)
    ok = np.isfinite(want)
    assert np.all(np.isfinite(got[ok])), what
    err = np.abs(got[ok] - want[ok])
    lim = rtol * np.abs(want[ok]) + 1e-300
    assert np.all(err <= lim), (what, float(np.max(err / np.maximum(np.abs(want[ok]), 1e-300))))
    return res


def golden_names():
    z = np.load(os.path.join(GOLD, "frames.npz"))
    return sorted(k[3:] for k in z.files if k.startswith("in_"))


@pytest.mark.parametrize("name", golden_names())
def test_contrast_is_an_exact_selection_on_the_goldens(plan, name):
    z = np.load(os.path.join(GOLD, "frames.npz"))
    x = z["in_" + name]
    res = check_against_own_magnitudes(plan, x, name)
    if name != "impulse":     # a flat spectrum: which bins are "lowest" is decided by the FFT's rounding noise
        a, b = FIELDS["sub_contrast"]
        # against the reference's own objects the magnitudes differ by the two FFTs' rounding (<= 1e-12 of the
        # frame's largest bin, i.e. up to ~1e-8 relative on leakage-floor bins), which the exponent amplifies
        _tol.check("sub_contrast", res["sub_contrast"].reshape(-1, 14), z["ref_" + name][:, a:b], 1e-7, 1e-12, what=name + " vs reference ")


def test_band_mean_next_to_one(plan):
    """Two bin-centred tones in the two-bin band (bins 1, 2) scaled so that the band mean is 1.0005: the exponent
    1 / ln(mean) is ~2000 and amplifies any error of the valley / peak means."""
    n = np.arange(2048 + 1024 * 3)
    x = np.sin(2 * np.pi * 1.0 * n / 2048.0) + np.sin(2 * np.pi * 2.0 * n / 2048.0 + 0.7)
    res = plan.extract([x], MASK)
    m = res["magnitude"][0, 1:3].mean()
    for target in (1.0005, 0.9995, 1.00001):
        y = x * (target / m)
        r = check_against_own_magnitudes(plan, y, f"mean {target}", rtol=1e-9)
        assert abs(r["magnitude"][0, 1:3].mean() - target) < 1e-6
    o = Oracle().run(x * (1.0005 / m))
    a, b = FIELDS["sub_contrast"]
    got = plan.extract([x * (1.0005 / m)], MASK)["sub_contrast"].reshape(-1, 14)
    _tol.check("sub_contrast", got[:, 0], o[:, a], 1e-6, 0.0, what="band mean 1.0005 vs oracle ")   # FFT rounding x 2000


def test_ties_in_the_sort_keys(plan):
    rng = np.random.default_rng(8)
    silence = np.zeros(2048 + 1024 * 2, np.float32)                       # every bin exactly 0: all keys equal
    impulse = np.zeros(2048 + 1024 * 2, np.float32); impulse[1500] = 0.8  # flat spectrum up to rounding: keys tie, doubles differ
    steps = np.repeat(rng.uniform(-1, 1, 40), 128)[:2048 + 1024 * 2].astype(np.float32)
    tone = (0.9 * np.sin(2 * np.pi * 3000.0 * np.arange(2048 + 1024 * 2) / 44100.0)).astype(np.float32)
    for name, x in (("silence", silence), ("impulse", impulse), ("steps", steps), ("tone", tone)):
        check_against_own_magnitudes(plan, x, name)


def test_quantised_spectrum_forces_the_slow_path(plan):
    """Magnitudes that agree to 19 mantissa bits but differ below: built in the time domain from a sum of
    bin-centred tones of nearly equal amplitude across one wide band."""
    n = np.arange(2048 + 1024)
    x = np.zeros(n.size)
    rng = np.random.default_rng(17)
    for k in range(470, 750):          # the 287-bin band (bins 465..751)
        x += 1e-3 * (1.0 + 1e-9 * rng.uniform(-1, 1)) * np.sin(2 * np.pi * k * n / 2048.0 + rng.uniform(0, 6.28))
    check_against_own_magnitudes(plan, x, "near-equal tones", rtol=1e-9)
